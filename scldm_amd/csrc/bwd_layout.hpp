// Unit counts of the fused backward layer's weight stream (packed by dit_aux.hpp: pack_bwd_val, consumed by dit_backward.hpp).
#pragma once
namespace scldm {
constexpr int kBwdChunk = 256;                // hidden units per backward chunk
constexpr int kBwdChunks = 3;                 // 684 -> 768
constexpr int kBwdUnitsChunk = 80;            // c_proj^T 16 | w1 16 | w2 16 | [w1^T | w2^T] 32
constexpr int kBwdUnitsLayer = kBwdChunks * kBwdUnitsChunk + 16 + 48 + 48;   // + c_proj(attn)^T 16 | c_attn 48 | c_attn^T 48 = 352 per wave
}  // namespace scldm
