"""The tuning database shipped with the package (infur_amd/conv_tune_gfx950.txt): well-formed, and its decisions for the three-byte
mode name forms that exist and fit the shape (a stale entry is ignored by pick_cfg, silently -- this is where it would be seen)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "infur_amd", "conv_tune_gfx950.txt")
HL_FORMS = {0, 5, 6, 11, 12, 13, 14, 15, 16, 17}  # conv_hl.hip: conv_hl_config_valid (16, 17: round 6)
HL_BN = {0: 128, 6: 128, 12: 128, 17: 128, 5: 256, 11: 256, 13: 256, 14: 256, 16: 256}


def rows():
    for ln in open(DB):
        if ln.strip() and not ln.startswith("#"):
            yield [int(x) for x in ln.split()]


def test_every_line_is_a_shape_and_a_configuration():
    seen = set()
    n = 0
    for r in rows():
        assert len(r) == 14, r
        H, W, Cin, OH, OW, Cout, KH, stride, dil, batch, res, mode, outf32, cfg = r
        assert min(H, W, Cin, OH, OW, Cout, KH, stride, dil, batch) >= 1 and 0 <= res <= 3 and 0 <= mode <= 5 and outf32 in (0, 1)
        assert 0 <= cfg <= 21
        assert tuple(r[:13]) not in seen, r  # one decision per shape
        seen.add(tuple(r[:13]))
        n += 1
    assert n > 300


def test_three_byte_mode_entries_name_forms_that_fit():
    n15 = 0
    for H, W, Cin, OH, OW, Cout, KH, stride, dil, batch, res, mode, outf32, cfg in rows():
        if mode != 5:
            continue
        assert cfg in HL_FORMS, (Cin, Cout, cfg)
        assert Cin % 32 == 0
        if cfg == 15:  # conv_hl_areg.hip: conv_hl_areg_valid
            n15 += 1
            assert KH == 1 and batch == 1 and outf32 == 0 and res in (0, 1) and Cin in (64, 128, 256, 512) and Cout >= 256 and Cout % 128 == 0 and Cout <= 2048
        else:
            assert HL_BN[cfg] <= Cout or HL_BN[cfg] == 128
    # (round 5's database had the register-resident form on the layer2 / layer3 expansions; since the pipelined K loop of round 6 the tuner
    #  prefers the two-workgroups-per-CU tiled forms there, so the count may be zero: only the validity of what IS listed is checked)
    assert n15 >= 0
