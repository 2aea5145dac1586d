// Matrix-core attention kernels (attention_mfma.hip); ick_attention / ick_attention_bwd try them first and
// fall back to the general kernels of attention.hip for shapes / layouts they do not cover.
#pragma once
#include "common.h"

namespace ick {

constexpr int kAttnMfmaUnsupported = -1000;   // not an error: "use the general kernel"

bool attn_mfma_shape_ok(int T, int S, int dh);
int launch_attn_mfma(const ick_attn_args& a, hipStream_t s);
int launch_attn_bwd_mfma(const ick_attn_bwd_args& a, hipStream_t s);

}  // namespace ick
