// Minimal ONNX (protobuf wire format) reader for FCN-ResNet model files -> INFURW01 blob.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace infur {

struct OnnxInfo {
    std::string input_name, input_dtype;
    std::vector<std::string> output_names;
    int depth = 0, num_classes = 0;
    bool aux = false;
    bool input_u8 = false, input_nhwc = false;  // the declared image input (predict_onnx.rs:223-265)
};

bool looks_like_onnx(const uint8_t* data, size_t len);
// 0 = ok; 1 = malformed / unsupported file; 2 = parsed but not a model this path can run
// (input-layout errors carry the reference's messages, predict_onnx.rs:228-262)
// a float model gives an INFURW01 blob, a QOperator int8 model (QLinearConv nodes) an INFURQ01 one
int onnx_to_blob(const uint8_t* data, size_t len, std::vector<uint8_t>& blob, OnnxInfo& info, std::string& err);
int onnx_q_to_blob(const uint8_t* data, size_t len, std::vector<uint8_t>& blob, OnnxInfo& info, std::string& err);  // onnx_qreader.cpp

}  // namespace infur
