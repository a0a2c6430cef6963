// Blob layout of the split-bf16 ("bf16x3") networks, shared by the host packer and the stage kernel (mlp_bf16x3.hip).
//
//   [ f32 section ]  the exact-f32 packed blob of the same network (stnerf_pack_net: SpaceLayout / MotionLayout) -- the
//                    per-ray prologue (mlp_raybias.hip) and the scalar head biases read it
//   [ consts      ]  what the kernel keeps in LDS for a whole work item, in its LDS order (fp32): bias vectors, head weights
//   [ stream      ]  every MFMA layer's weights as bf16 triples in the exact order the K loops consume them, in 24 KB
//                    slots (= 2 K steps of a 128-feature pass): [K step][feature block of 32][piece][lane][8 bf16]
#pragma once
#include "mlp_common.h"

namespace stnerf {

constexpr int BX_CHUNK = 1024;                 // one wave-wide 16-byte access
constexpr int BX_UNIT = 3 * BX_CHUNK;          // the three pieces of one (K step of 16, 32 output features) A operand
constexpr int BX_SLOT = 8 * BX_UNIT;           // 2 K steps x 4 feature blocks
constexpr int BX_RING = 4;                     // slots of the LDS ring
constexpr int BX_CONST_SPACE = 3072;           // floats
constexpr int BX_CONST_MOTION = 1024;
// SpaceNet consts (float offsets): b[i] (stage1.0 .. stage2.4), the deep_rgb biases, density_net.0's row, the colour head
constexpr int BXC_B = 0, BXC_B_DEEP = 1792, BXC_W_SIGMA = 2048, BXC_W_RGB2 = 2304;
// MotionNet consts: b[i] (motion_net.0 .. .8), the flow head
constexpr int BXM_B = 0, BXM_W_OUT = 640;

struct BxLayout {
    int64_t f32_floats;   // length of the f32 section
    int64_t consts_off;   // bytes
    int64_t stream_off;   // bytes
    int32_t n_slots;
    int64_t total_bytes;
};

__host__ __device__ inline int bx_space_slots(bool deep) {
    // stage1.0: 2 passes x 4 K steps; five 256-wide layers: 2 x 16; stage2.0: 2 x 20; rgb_net.1: 16; deep: 2 x 8
    return (2 * 4 + 5 * 2 * 16 + 2 * 20 + 16 + (deep ? 16 : 0)) / 2;
}
__host__ __device__ inline int bx_motion_slots() { return (6 + 4 * 8) / 2; }

__host__ __device__ inline BxLayout bx_layout(int kind) {
    BxLayout B;
    const bool space = STNERF_NET_IS_SPACE(kind);
    B.f32_floats = space ? space_layout(STNERF_NET_USES_TIME(kind), STNERF_NET_IS_DEEP(kind)).total : motion_layout().total;
    B.consts_off = (B.f32_floats * 4 + 1023) & ~int64_t(1023);
    B.stream_off = B.consts_off + (space ? BX_CONST_SPACE : BX_CONST_MOTION) * 4;
    B.n_slots = space ? bx_space_slots(STNERF_NET_IS_DEEP(kind)) : bx_motion_slots();
    B.total_bytes = B.stream_off + (int64_t)B.n_slots * BX_SLOT;
    return B;
}

// Input column of B-operand position (K step t, lane half h, element j):
//   layers fed by the previous layer's accumulators: register 8 (t & 1) + j of block t >> 1 = feature 32 fb + 8 q + 4 h + r
__host__ __device__ inline int bx_kmap_hidden(int t, int h, int j) { return 32 * (t >> 1) + 8 * (2 * (t & 1) + (j >> 2)) + 4 * h + (j & 3); }
//   layers fed by a staged encoding: 8 consecutive features per lane half
__host__ __device__ inline int bx_kmap_enc(int t, int h, int j) { return 16 * t + 8 * h + j; }

// mlp_bf16x3.hip
struct StageArgs;
struct StoreTapArgs;
int launch_bf16x3_stage(const StageArgs& a, bool deep_rgb, int cus, hipStream_t stream);
int launch_bf16x3_stage_store(const StageArgs& a, const StoreTapArgs& t, int cus, hipStream_t stream);

}  // namespace stnerf
