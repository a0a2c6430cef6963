// Argument block and row bookkeeping shared by the persistent stage kernels behind stnerf_mlp_stage (launched from
// stage_entry.hip): mlp_wave.hip (exact f32) and mlp_bf16x3.hip (split bf16) -- sample-split waves, activations in registers;
// their common prologue / epilogue code is mlp_wave_common.h.  (The round-2 organisation with feature-split waves and
// activations in LDS, mlp_stage.hip, was removed in round 3.)
#pragma once
#include "mlp_blocks.h"

namespace stnerf {

struct StageLayer {
    const float* space;        // packed SpaceNet
    const float* motion;       // packed MotionNet or nullptr
    const int32_t* ray_list;   // compacted hit rays or nullptr (every ray)
    const int32_t* ray_count;
    const float* xyz;          // this layer's sample points (ray stride StageArgs::xyz_ray_stride)
    float* raw;                // this layer's {r,g,b,sigma} (ray stride StageArgs::raw_ray_stride)
    const float* times;        // this layer's frame-id column or nullptr
    int32_t use_time;          // the SpaceNet takes the time encoding
    int32_t motion_flags;      // STNERF_MOTION_PLAIN_TIME
    const float* raybias;      // [n_rays][128]: C operands of rgb_net.1 per ray (mlp_raybias.hip)
};

struct StageArgs {
    StageLayer layer[STNERF_MAX_LAYERS];
    int32_t n_layers, ns;
    int64_t n_rays;
    int64_t xyz_ray_stride, raw_ray_stride, dirs_ray_stride, times_ray_stride;
    const float* dirs;
    uint32_t* queue;           // one counter, zero at launch
    int32_t sigmoid_rgb;       // store sigmoid(rgb) (layers/render_layer.py:47) instead of the raw colour output
    // development builds of mlp_wave.hip (-DSTNERF_WAVE_DEBUG): activations of queue slot 0 after stage `dbg_stage` go to
    // dbg[row][256] (tools/ab_wave.py localises a wrong layer with them); null otherwise
    float* dbg;
    int32_t dbg_stage;
};

// rows of a layer = hit rays x samples per ray
__device__ __forceinline__ int64_t layer_rows(const StageLayer& ly, int64_t n_rays, int ns) {
    int64_t cnt = n_rays;
    if (ly.ray_count) {
        const int64_t c = *ly.ray_count;
        cnt = c < cnt ? c : cnt;
    }
    return cnt * ns;
}

struct RowRef {
    int64_t ray;
    int k;
    bool valid;
};
__device__ __forceinline__ RowRef locate_row(const int32_t* ray_list, int64_t row, int64_t rows, int ns) {
    RowRef r{0, 0, row < rows};
    if (r.valid) {
        const int64_t slot = row / ns;
        r.k = (int)(row - slot * ns);
        r.ray = ray_list ? (int64_t)ray_list[slot] : slot;
    }
    return r;
}

// The fractional-time lerp of the MotionNet encodings (modeling/motion_net.py:49-60); one definition so that every
// kernel evaluates the same expression.
__device__ __forceinline__ float lerp_enc(bool frac, float om, float wgt, float va, float vb) {
    return frac ? om * va + wgt * vb : va;
}

// Training: where a stage kernel's tap writes every layer's input (mlp_wave.hip: StoreTap, exact f32; mlp_bf16x3.hip:
// BxStoreTap, split bf16).  Row r of the launch <-> row r of every matrix.
struct NoTapArgs {};
constexpr int TAP_PE = 100;   // the taps' stage id of the staged encoding
struct StoreTapArgs {
    float* buf[8];     // stage s: the input of stage1.2 .. stage2.4 (0 .. 5), of the heads / rgb_net.1 (6: 256 wide), of rgb_net.3 (7: 128)
    int32_t ld[8];     // row strides in floats (multiples of 4; 16-byte aligned bases)
    float* pe;         // PE(pos): 63 features + one zero
    int32_t ld_pe;
    uint32_t* bits;    // [8][.. bits_stride ..]: rows x 8 words per stage: the ReLU masks of stages 0 .. 7 as bit planes (see StoreTap), or null
    int64_t bits_stride;   // uint32 words between two stages' planes
};

// mlp_wave.hip
int launch_wave_stage(const StageArgs& a, bool deep_rgb, int cus, hipStream_t stream);
int launch_wave_stage_store(const StageArgs& a, float* const (&buf)[8], const int32_t (&ld)[8], float* pe, int32_t ld_pe, uint32_t* bits,
                            int64_t bits_stride, int cus, hipStream_t stream);

}  // namespace stnerf
