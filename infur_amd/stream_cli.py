"""Headless streaming front end: packed bgr24 frames in, RGBA masks out.

    ffmpeg -i in.mp4 -an -f image2pipe -fflags nobuffer -pix_fmt bgr24 -c:v rawvideo pipe:1 \
      | python -m infur_amd.stream_cli --width 1280 --height 720 --scale 0.5 --model fcn.infurw > masks.rgba

stdin carries exactly what the reference's decoder reads from its ffmpeg child
(ff-video/src/decoder.rs:53-64,156-165): W*H*3 bytes per frame.  stdout receives one
premultiplied RGBA mask (ow*oh*4 bytes) per frame, in order.  A ring of ``--depth`` frames is in
flight (the reference's bounded channel of 2, infur/src/main.rs:105): uploads, kernels and
downloads of neighbouring frames overlap.  ``--model`` takes an INFURW01 blob or a float
fcn-resnet50/101 .onnx file; ``--synthetic-weights`` uses the seeded test weights instead.
"""
from __future__ import annotations

import argparse
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--bilinear", action="store_true", help="bilinear Scale instead of the reference's nearest")
    ap.add_argument("--model", default="")
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32s", "f32x", "f16hl", "f16"],
                    help="f32: exact f32 MFMA; f32s: f32 tensors on the f16 matrix cores (hi+lo pairs), same logits, ~1.9x; f32x: f32s with the cross terms on the fp8 MFMA, logits 1.5e-4, ~2.1x; "
                         "f16hl: three-byte tensors, logits 1.5e-4, ~2.5x; f16: logits 2e-3, ~4x")
    ap.add_argument("--copy", action="store_true", help="the copying submit / collect calls instead of reading the pipe straight into the ring's pinned slots")
    ap.add_argument("--depth", type=int, default=2, help="frames in flight")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--input", default="-", help="raw bgr24 file (default stdin)")
    ap.add_argument("--output", default="-", help="raw rgba file (default stdout)")
    a = ap.parse_args(argv)

    from . import _lib
    from .app import RawVideoSource, StreamPath, VideoProcError
    from .processors import Context, Model, ModelCmd

    ctx = Context(device=a.device, dtype=a.dtype)
    model = Model(ctx)
    if a.synthetic_weights:
        from .weights import synth_blob

        model.control(ModelCmd.LoadBlob(synth_blob()))
    elif a.model:
        model.control(ModelCmd.Load(a.model))
    else:
        ap.error("give --model PATH or --synthetic-weights")

    fin = sys.stdin.buffer if a.input == "-" else open(a.input, "rb")
    fout = sys.stdout.buffer if a.output == "-" else open(a.output, "wb")
    src = RawVideoSource(fin, a.width, a.height, close_stream=a.input != "-")
    sp = StreamPath(ctx, depth=a.depth, scale_mode=_lib.SCALE_BILINEAR if a.bilinear else _lib.SCALE_NEAREST)

    n, t0 = 0, time.perf_counter()
    if a.copy:
        def frames():
            img = src.empty_image()
            while True:
                try:
                    fid = src.read_frame(img)
                except VideoProcError as e:
                    if e.kind == "FinishedNormally":
                        return
                    raise
                yield fid, img  # submit() copies the frame into a pinned slot before returning

        for _fid, rgba in sp.run(frames(), a.scale):
            fout.write(memoryview(rgba).cast("B"))
            n += 1
    else:
        # zero-copy: the pipe is read straight into the next pinned slot (the reference's decoder fills its reused BgrImage the same
        # way, ff-video/src/decoder.rs:156-165) and the mask is written out of the pinned output slot
        def drain_one():
            _fid, rgba, _ = sp.collect_view()
            fout.write(memoryview(rgba).cast("B"))
            sp.release()

        while True:
            if sp.pending() >= a.depth:
                drain_one()
                n += 1
            slot = sp.acquire(a.width, a.height, a.scale)
            try:
                fid = src.read_frame(slot)
            except VideoProcError as e:
                sp.abandon()  # the acquired slot goes back uncommitted
                if e.kind == "FinishedNormally":
                    break
                raise
            sp.commit(a.width, a.height, a.scale, fid)
        while sp.pending():
            drain_one()
            n += 1
    fout.flush()
    el = time.perf_counter() - t0
    sys.stderr.write(f"infur stream: {n} frames in {el:.2f} s ({n / max(el, 1e-9):.1f} frames/s)\n")
    sp.close()
    ctx.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
