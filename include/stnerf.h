/*
 * stnerf.h -- C ABI of libstnerf_hip.so: the MI355X (gfx950) implementation of the st-nerf layered
 * ray-march hot path.
 *
 * The reference (DarlingHang/st-nerf) has no FFI: its boundary for this path is the Python call
 * surface of LayeredRFRender.forward / layered_batchify_ray and the ops they compose.  Each entry
 * point below replaces one of those ops (cited as reference file:line, relative to
 * /root/reference) and is what a ctypes / cffi / pybind binding in the reference would bind
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - Plain C: pointers + sizes.  Every `const float*` / `float*` / `uint8_t*` / `int32_t*` argument
 *     is a DEVICE pointer unless its comment says "host".  All tensors are dense row-major fp32.
 *   - The caller owns every buffer (inputs, outputs, packed weights).  The library allocates nothing
 *     on the device and never frees caller memory.
 *   - All work is enqueued on the caller's stream (`stream` = a hipStream_t, may be NULL for the
 *     default stream).  No entry point synchronises the device or the stream.
 *   - Return value: 0 on success, a negative STNERF_E* code otherwise; stnerf_last_error() gives
 *     the message (thread-local).  Nothing throws or aborts across the ABI (the reference calls
 *     exit(-1) on a bad ray layout, modeling/layered_rfrender.py:161-163; here that is -EINVAL).
 *   - Layouts used by the chunk pipeline ("ray-major"): t[n][l][S], xyz[n][l][S][3],
 *     raw[n][l][S][4] = {r, g, b, sigma} (network outputs before any activation), mask[n][l].
 */
#ifndef STNERF_H
#define STNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STNERF_OK 0
#define STNERF_EINVAL (-1)   /* bad argument / shape */
#define STNERF_ELAUNCH (-2)  /* HIP launch or runtime error */
#define STNERF_EARCH (-3)    /* device is not gfx950 */

#define STNERF_MAX_LAYERS 16

typedef void* stnerf_stream_t; /* hipStream_t */

/* library / device ----------------------------------------------------------------------- */
const char* stnerf_version(void);
const char* stnerf_last_error(void);
/* cu_count, lds_bytes_per_cu, clock_khz may be NULL; arch receives e.g. "gfx950" (host buffer). */
int stnerf_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* arch, int arch_len);

/* Launch profiler: between _begin and _end every kernel launch of the entry points below is bracketed by a
 * HIP event pair recorded on the launch stream.  _end synchronises those events only and returns one record per
 * launch, in launch order (n_records = number of launches, even if larger than max_records).
 * kernel: 0 spacenet, 1 motionnet, 2 composite, 3 resample, 4 sample_coarse; kind: the net kind for 0/1;
 * n_rays x ns = the launch's upper bound on rows (masked launches process ray_count x ns of them);
 * tag: the layer a stnerf_render_rays launch belongs to (-1 otherwise); bytes_per_ray: algorithmic HBM bytes
 * per ray for the HBM-bound kernels (2..4), 0 for the networks. */
typedef struct stnerf_profile_record {
    int32_t kernel, kind, ns, tag;
    int64_t n_rays, bytes_per_ray;
    float ms;
    int32_t pad_;
} stnerf_profile_record;
int stnerf_profile_begin(void);
int stnerf_profile_end(stnerf_profile_record* records_host, int max_records, int* n_records);

/* Per-layer edit applied to sample points (inverse of the box edit), host struct passed by value
 * inside stnerf_scene.  modeling/layered_rfrender.py:293-303 (coarse) and :467-475 (fine). */
typedef struct stnerf_layer_edit {
    float shift[3];  /* x -= shift              (only if has_shift) */
    float scale;     /* x = (x - pivot) / scale + pivot   (only if has_scale) */
    int32_t has_shift;
    int32_t has_scale;
} stnerf_layer_edit;

/* Which rays of a view a call works on.  Local ray i of a call is GLOBAL ray
 *     first + i                                    (stripe == 0: one contiguous window)
 *     first + (i / stripe) * period + i % stripe   (stripe  > 0: stripes of `stripe` rays, one every `period` rays --
 *                                                   what rank r of G takes of a view split into interleaved stripes:
 *                                                   first = r * stripe, period = G * stripe)
 * The global index selects the pixel (stnerf_generate_rays) and keys the device RNG (stnerf_sample_coarse,
 * stnerf_resample, stnerf_render_rays), so an image does not depend on how its rays are cut into launches, chunks
 * or GPU shards.  The three numbers travel as (first_ray | ray_index_base, ray_index_stripe, ray_index_period). */

/* a1 + a2: pinhole rays of the n rays of the window (first_ray, stripe, period) of an h x w view, row-major over
 * (row, col), written as [origin(3), dir(3), frame_ids(n_frame_cols)] with `ray_stride` floats per
 * ray.  Replaces utils/render_helpers.py:42-128 (generate_rays), utils/ray_sampling.py:22-72 and
 * data/datasets/ray_dataset.py:276-281.  Kinv = inverse intrinsics (host, 9 floats, row-major),
 * T = camera-to-world (host, 16 floats), frame_ids host array of n_frame_cols floats (or NULL). */
int stnerf_generate_rays(const float* Kinv_host, const float* T_host, int h, int w, int64_t first_ray,
                         int64_t ray_index_stripe, int64_t ray_index_period, int64_t n,
                         const float* frame_ids_host, int n_frame_cols, float* rays, int ray_stride,
                         stnerf_stream_t stream);

/* a5: ray / 8-corner-box slab test.  layers/RaySamplePoint.py:8-62 (intersection).
 * boxes: [l][8][3] shared by all rays (box_ray_stride = 0) or per ray (box_ray_stride = l*24).
 * far_near out: [n][l][2] = (far, near), (-1000,-1000) on a miss. */
int stnerf_intersect(const float* rays, int64_t n, int ray_stride, const float* boxes,
                     int64_t box_ray_stride, int l, float* far_near, stnerf_stream_t stream);

/* a5 + a6 (+ the point un-edit of a4): stratified jittered coarse samples of every layer.
 * layers/RaySamplePoint.py:70-107 (RaySamplePoint.forward).
 * jitter: [l][n][n1] uniform draws to REPLAY (the reference's torch.rand tensors), or NULL to draw
 * on the device: Philox4x32-10 keyed by (seed, global ray index, layer, sample) -- independent
 * of chunking (global index: see the ray-window note above).  edits: host array of l entries or NULL; pivot: host, 3 floats.
 * Outputs: t[n][l][n1], xyz[n][l][n1][3] (may be NULL), mask[n][l]: bit 0 = the reference's ray_mask (|bin width| > 1e-5,
 * :105); bit 1 = a HINT for stnerf_composite: the ray misses the layer's box altogether (both slab hits -1000, :53-62), so
 * every depth of the (ray, layer) pair is exactly -1000 and the compositor need not read them to find that out (40 % of a
 * nine-layer ray's depth bytes).  Consumers of the mask as the reference's boolean take `mask & 1`; stnerf_compact_rays and
 * stnerf_composite do.  A mask without hints (bit 1 clear everywhere) is always valid; a caller-made mask must be strictly 0 / 1
 * (any other value is read as hit-bit + hint).  stnerf_render_rays clears the hints before it returns: its `mask` output is 0 / 1. */
int stnerf_sample_coarse(const float* rays, int64_t n, int ray_stride, const float* boxes,
                         int64_t box_ray_stride, int l, int n1, const float* jitter, uint64_t seed,
                         int64_t ray_index_base, int64_t ray_index_stripe, int64_t ray_index_period,
                         const stnerf_layer_edit* edits_host, const float* pivot_host, float* t, float* xyz,
                         uint8_t* mask, stnerf_stream_t stream);

/* Ragged work: list of rays whose mask[ray][layer] is set.  Replaces the boolean-mask indexing
 * (and its host sync) at modeling/layered_rfrender.py:344-353,400-413,497-510,555-563.
 * ray_list[layer][n] (int32; row `layer` gets the hits, in no particular order), ray_count[layer].
 * ray_count must be zeroed by the caller before the call (it is accumulated with atomics). */
int stnerf_compact_rays(const uint8_t* mask, int64_t n, int l, int32_t* ray_list, int32_t* ray_count,
                        stnerf_stream_t stream);

/* Network weights ------------------------------------------------------------------------- */
#define STNERF_NET_SPACE 0      /* SpaceNet(use_time=False)  modeling/spacenet.py:16-86  */
#define STNERF_NET_SPACE_TIME 1 /* SpaceNet(use_time=True)                               */
#define STNERF_NET_MOTION 2     /* MotionNet(c_input=4)  modeling/motion_net.py:7-32     */
#define STNERF_NET_SPACE_DEEP 3      /* SpaceNet(use_time=False, deep_rgb=True)  modeling/spacenet.py:68-79 */
#define STNERF_NET_SPACE_TIME_DEEP 4 /* SpaceNet(use_time=True,  deep_rgb=True)                             */
#define STNERF_NET_IS_SPACE(kind) ((kind) == STNERF_NET_SPACE || (kind) == STNERF_NET_SPACE_TIME || \
                                   (kind) == STNERF_NET_SPACE_DEEP || (kind) == STNERF_NET_SPACE_TIME_DEEP)
#define STNERF_NET_USES_TIME(kind) ((kind) == STNERF_NET_SPACE_TIME || (kind) == STNERF_NET_SPACE_TIME_DEEP)
#define STNERF_NET_IS_DEEP(kind) ((kind) == STNERF_NET_SPACE_DEEP || (kind) == STNERF_NET_SPACE_TIME_DEEP)

/* Bytes of the packed (kernel-layout) weight blob of one network. */
int64_t stnerf_packed_bytes(int kind);
/* stnerf_pack_net with every tensor and the destination IN DEVICE MEMORY (weights_dev / biases_dev: host arrays of device pointers;
 * same tensors, same blob, bit for bit): one memset + one kernel on `stream`, no host round trip -- what a training loop calls after
 * every optimizer.step() (round 5). */
int stnerf_pack_net_device(int kind, const float* const* weights_dev, const float* const* biases_dev, int n_tensors, void* dst_dev,
                           int64_t dst_bytes, stnerf_stream_t stream);
/* The A operands of the fused backward chains (stnerf_train_spacenet_dx / stnerf_train_motionnet_dx): `count` (<= 12) sections of one
 * destination blob, each an nn.Linear weight W (n_out x n_in, row stride ldw floats, device memory) as [n_out / 4][n_pad][4] --
 * dst[dst_off + ((o / 4) n_pad + n) 4 + o % 4] = W[o][n], zero for n_in <= n < n_pad -- or, n_pad == 0, a plain copy of its n_out x n_in
 * floats (the heads).  One launch, no host round trip: a training loop rebuilds them after every optimizer.step(). */
typedef struct stnerf_transpose_section {
    const float* w;
    int64_t ldw;
    int64_t dst_off;   /* floats */
    int32_t n_out, n_in, n_pad;
} stnerf_transpose_section;
int stnerf_pack_transposed(const stnerf_transpose_section* sections, int count, float* dst_dev, int64_t dst_floats, stnerf_stream_t stream);
/* Repack reference-layout tensors (nn.Linear: weight (out,in) row-major, bias (out)) into the
 * kernel layout.  HOST -> HOST; the caller uploads `dst` to the device (any allocator).
 * SpaceNet order (10 tensors): stage1.{0,2,4,6}, stage2.{0,2,4}, density_net.0, rgb_net.{1,3};
 * deep_rgb kinds (12 tensors): ..., density_net.0, rgb_net.{1,3,5,7}.
 * MotionNet order (6): motion_net.{0,2,4,6,8,10}.  Checkpoint key names: SURVEY.md section 5. */
int stnerf_pack_net(int kind, const float* const* weights_host, const float* const* biases_host,
                    int n_tensors, void* dst_host, int64_t dst_bytes);

/* a7 + a9: fused positional encoding + SpaceNet MLP (fp32 MFMA).  modeling/spacenet.py:101-160.
 * Work list: ray slot s in [0, count) -> ray j = ray_list ? ray_list[s] : s, where count =
 * ray_count ? min(*ray_count, n_rays) : n_rays; every ray contributes ns samples.
 *   pos   of (j,k): xyz  + j*xyz_ray_stride  + 3k      (3 floats)
 *   dir   of  j   : dirs + j*dirs_ray_stride           (3 floats)
 *   time  of  j   : times + j*times_ray_stride         (1 float; kinds with time only)
 *   out   of (j,k): raw  + j*raw_ray_stride  + 4k      = {r, g, b, sigma}, no activation
 * All strides in floats. */
int stnerf_spacenet_fwd(int kind, const void* packed, int64_t n_rays, int ns, const int32_t* ray_list,
                        const int32_t* ray_count, const float* xyz, int64_t xyz_ray_stride,
                        const float* dirs, int64_t dirs_ray_stride, const float* times,
                        int64_t times_ray_stride, float* raw, int64_t raw_ray_stride, float* ray_bias,
                        stnerf_stream_t stream);

/* The per-ray part of rgb_net.1 (modeling/spacenet.py:80-86,141-151): the layer reads the 256 backbone features of a
 * sample and the encodings of the ray's direction and frame id -- the same 27 (+ 21) numbers for every sample of the ray
 * (:115,118 repeat them).  out[j][0..127] = bias + W[:, 256:] * relu([PE_4(dir_j), PE_10(time_j)]) for the listed rays j
 * (rows of other rays are left untouched); the exact-f32 MLP kernels take row j as the C operand of the layer's first
 * MFMA.  stnerf_spacenet_fwd and stnerf_mlp_stage call this themselves into their `ray_bias` workspaces
 * ([n_rays][128] floats per layer, 16-byte aligned); exported for tests and for callers that want the table. */
int stnerf_rgb_ray_bias(int kind, const void* packed, int64_t n_rays, const int32_t* ray_list, const int32_t* ray_count,
                        const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride,
                        float* out, stnerf_stream_t stream);

/* a7 + a8: fused positional encoding (with the fractional-time lerp) + MotionNet MLP.
 * modeling/motion_net.py:34-71.  Same work list as above.  flow (may be NULL) gets the 3-vector at
 * flow + j*flow_ray_stride + 3k.  `add_to_xyz` is a set of STNERF_MOTION_* bits: ADD_TO_XYZ updates the
 * point in place (xyz += flow), which is what modeling/layered_rfrender.py:355-356,509-510 do; PLAIN_TIME
 * selects MotionNet(input_time=False) (motion_net.py:61-62: PE of [xyz, t] as given, no floor/ceil lerp),
 * the flavour of bkgd_time_deform_net (layered_rfrender.py:92-93). */
#define STNERF_MOTION_ADD_TO_XYZ 1
#define STNERF_MOTION_PLAIN_TIME 2
int stnerf_motionnet_fwd(const void* packed, int64_t n_rays, int ns, const int32_t* ray_list,
                         const int32_t* ray_count, float* xyz, int64_t xyz_ray_stride,
                         const float* times, int64_t times_ray_stride, float* flow,
                         int64_t flow_ray_stride, int add_to_xyz, stnerf_stream_t stream);

/* "bf16x3": the packed form of a network for the split-bf16 arithmetic of stnerf_mlp_stage (STNERF_STAGE_BF16X3;
 * csrc/mlp_bf16x3.hip).  modeling/spacenet.py:45-86, modeling/motion_net.py:20-32 are plain fp32 nn.Linear layers: every
 * weight (host side, here) and every activation (in the kernel) is split into three bf16 numbers, x = x0 + x1 + x2 --
 * 8 + 8 + 8 significand bits, i.e. exact for every finite fp32 value inside bf16's exponent range (|x| <= 3.39e38: above
 * bf16's largest finite value the leading piece rounds to inf; residual pieces below 2^-133 flush, an error < 2^-16 of
 * the smallest normal number), with fp32's exponent range: no |W| limit, no activation limit, no overflow flag -- and a
 * product is evaluated with its six leading cross terms on the bf16 MFMA (dropped terms <= 2^-24 |a b|),
 * fp32 accumulate.  Same tensors as stnerf_pack_net; the blob = [the exact-f32 blob of stnerf_pack_net | bias vectors and
 * head weights in the kernel's LDS order | the MFMA layers' weights as bf16 triples in consumption order].  All net
 * kinds.  The device copy must be 1 KB aligned.
 * Edge semantics (tests/test_gpu_stage.py::test_bf16x3_edge_semantics, tests/test_bf16x3_pack_cpu.py):
 *   weights / biases: STNERF_EINVAL for NaN, +-inf and |w| > 3.3895e38 (bf16's largest finite value; fp32's top 0.4 % cannot be split);
 *   fp32-subnormal weights are accepted, pieces below 2^-133 flush (absolute error < 2^-133 per weight);
 *   sample points must be FINITE (the sampler's and the resampler's are).  Neither stage kernel propagates NaN / inf the way ATen does:
 *   their ReLU is an integer max on the bit pattern (one instruction; -inf and sign-bit NaNs become 0), and the bf16 split turns what is
 *   left into zero pieces.  A sample with a NaN / inf coordinate therefore gets, in bf16x3, the FINITE outputs of a network whose first
 *   layer's activations are zero (the biases' response: the same values for every such sample) where ATen returns NaN; the exact-f32
 *   kernel returns NaN for NaN / +inf coordinates and those finite values for -inf.  Likewise activations that overflow fp32 (|x| > 3.4e38: ATen carries +-inf / NaN on) give unspecified
 *   values for that sample.  In every case ONLY that sample is affected: every other sample of the launch, of the same wave included,
 *   is bit-identical to a launch without it.  Subnormal activations and products behave as in the exact-f32 kernel up to the flush. */
int64_t stnerf_packed_bytes_bf16x3(int kind);
int stnerf_pack_net_bf16x3(int kind, const float* const* weights_host, const float* const* biases_host,
                           int n_tensors, void* dst_host, int64_t dst_bytes);
/* The same blob, bit for bit, from tensors IN DEVICE MEMORY into a 1 KB-aligned device destination (host arrays of device pointers;
 * one memset + two kernels on `stream`): what a training loop calls after every optimizer.step() when its forward runs in split
 * bf16 (stnerf_train_spacenet_fwd_bf16x3; round 6).  No finiteness check (no host round trip): a weight stnerf_pack_net_bf16x3 would
 * refuse gives NaN pieces and NaN outputs. */
int stnerf_pack_net_bf16x3_device(int kind, const float* const* weights_dev, const float* const* biases_dev, int n_tensors,
                                  void* dst_dev, int64_t dst_bytes, stnerf_stream_t stream);

/* a8 + a9 for a whole network stage of the pipeline (coarse or fine, modeling/layered_rfrender.py:340-418 / :495-576):
 * ONE persistent launch evaluates every listed layer -- one workgroup per CU pops work items (128 rows of a layer; the
 * MotionNet of a deformed layer runs in front of its SpaceNet on the same rows, the flow stays on chip) from a device-side
 * queue (csrc/stage_entry.hip).  A wave owns 32 samples and the activations never leave its registers.  Two arithmetics:
 * exact f32 (csrc/mlp_wave.hip, the default; results bit-identical to stnerf_motionnet_fwd(ADD_TO_XYZ) followed by
 * stnerf_spacenet_fwd per layer, except that the deformed points are NOT written back to xyz) and, with
 * STNERF_STAGE_BF16X3, split-bf16 (csrc/mlp_bf16x3.hip: every fp32 operand as three bf16 pieces, six bf16 MFMAs per
 * product, two fp32 accumulators -- fp32's exponent range, no range limits, closer to an fp64 evaluation than an fp32
 * fma chain is; every net must then be a blob of stnerf_pack_net_bf16x3, 1 KB aligned).  `queue`: one zeroed uint32 on
 * the device.  `ray_bias`: workspace of n_layers x n_rays x 128 floats (16-byte aligned) for the per-ray part of
 * rgb_net.1 (stnerf_rgb_ray_bias).
 * flags: STNERF_STAGE_DEEP_RGB = the SpaceNets are of a *_DEEP kind; STNERF_STAGE_SIGMOID_RGB = store sigmoid(rgb)
 * instead of the raw colour head output (torch.sigmoid of layers/render_layer.py:47 moved into the network epilogue,
 * where it is free; pair it with stnerf_composite_params.rgb_activated = 1); STNERF_STAGE_BF16X3 as above. */
#define STNERF_STAGE_DEEP_RGB 1
#define STNERF_STAGE_SIGMOID_RGB 2
#define STNERF_STAGE_BF16X3 4
typedef struct stnerf_stage_layer {
    const void* space;         /* packed SpaceNet (kind given by use_time and the deep_rgb argument)                 */
    const void* motion;        /* packed MotionNet evaluated first on the same rows (xyz + flow), or NULL            */
    const int32_t* ray_list;   /* compacted hit rays of the layer (stnerf_compact_rays) or NULL = every ray          */
    const int32_t* ray_count;
    const float* xyz;          /* [n][ns][3] with ray stride xyz_ray_stride                                         */
    float* raw;                /* [n][ns][4] with ray stride raw_ray_stride                                         */
    const float* times;        /* frame id of ray j at times[j * times_ray_stride]; NULL if neither net takes it    */
    int32_t use_time;          /* the SpaceNet is of a *_TIME kind                                                  */
    int32_t motion_flags;      /* STNERF_MOTION_PLAIN_TIME                                                          */
} stnerf_stage_layer;
int stnerf_mlp_stage(const stnerf_stage_layer* layers_host, int n_layers, int64_t n_rays, int ns, const float* dirs,
                     int64_t dirs_ray_stride, int64_t times_ray_stride, int64_t xyz_ray_stride,
                     int64_t raw_ray_stride, int flags, uint32_t* queue, float* ray_bias, stnerf_stream_t stream);

/* a7 standalone: NeRF positional encoding [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)],
 * each block `dim` wide.  utils/dimension_kernel.py:3-73 (Trigonometric_kernel.__call__).  In the render
 * path the encoding is fused into the MLP kernels; this entry point serves the op-level API.
 * x[n][dim] -> y[n][dim*(include_input + 2*n_freq)].  Same < 1 ulp sin/cos as the fused kernels. */
int stnerf_encode(const float* x, int64_t n, int dim, int n_freq, int include_input, float* y,
                  stnerf_stream_t stream);

/* a12 standalone: gen_weight(sigma, delta), layers/render_layer.py:8-17.  sigma[n][S] raw, delta[n][S]
 * -> weights[n][S] = alpha * exclusive cumprod(1 - alpha + 1e-10), alpha = 1 - exp(-relu(sigma) delta). */
int stnerf_gen_weight(const float* sigma, const float* delta, int64_t n, int S, float* weights,
                      stnerf_stream_t stream);

/* a10 + a11 + a12: post-network density edits, per-layer composite, cross-layer merge by depth
 * and merged composite, one pass per ray.  layers/render_layer.py:8-58 (gen_weight,
 * VolumeRenderer.forward); modeling/layered_rfrender.py:414-448 (coarse), :538-606 (fine). */
typedef struct stnerf_composite_params {
    float border;                              /* last interval, BOARDER_WEIGHT            */
    float near;                                /* model.near                               */
    int32_t fine;                              /* 0: coarse-stage rules, 1: fine-stage     */
    int32_t cut_negative_t;                    /* coarse: performer sigma[t<0]=0  (:414)   */
    float threshold[STNERF_MAX_LAYERS];        /* sigma < threshold -> 0 (retiming)        */
    int32_t use_threshold[STNERF_MAX_LAYERS];
    float sigma_scale[STNERF_MAX_LAYERS];      /* fine: layer 2 *= alpha (:575-576), else 1 */
    int32_t rgb_activated;                     /* 1: raw[..][0..2] already hold sigmoid(rgb) (stnerf_mlp_stage with
                                                * STNERF_STAGE_SIGMOID_RGB applies it in the network epilogue);
                                                * 0: raw network output, torch.sigmoid is applied here (render_layer.py:47) */
    int32_t evaluated[STNERF_MAX_LAYERS];      /* 0: layer's nets were skipped (hidden): raw is not read;
                                                * 1: evaluated on the rays `mask` marks (performers, :397-413);
                                                * 2: evaluated on EVERY ray, `mask` is not consulted (the background:
                                                *    bkgd_spacenet runs on all rays, :382-392, and is composited even
                                                *    where ray_mask[0] is False -- a ray through an edge of its box) */
} stnerf_composite_params;

/* t[n][l][S], raw[n][l][S][4], mask[n][l] (NULL = all set; a clear bit means "not evaluated":
 * rgb = sigma = 0 as the reference's zero tensors, :398-399).
 * Outputs (any may be NULL): layer_out[n][l][5] = {color(3), depth, acc}; mixed_out[n][5];
 * weights[n][l][S] per-layer weights (needed by stnerf_resample); order[n][l*S] int32 = source
 * index (layer*S + k) of each merged sample (torch.sort's index, ties broken by source index).
 * scratch: n bytes of device memory or NULL (contents on return unspecified).  With scratch the rays with a single live
 * layer (about half of a view) are composited first by a latency-pipelined kernel that needs no LDS, the others by the
 * register / insertion-merge kernel -- in two launches when a merged list of all l layers would not fit the LDS at full
 * occupancy (e.g. 9 x 192 samples); without scratch that kernel takes every ray in one launch.  The `order` output and
 * layers of more than 192 samples take the LDS-staged kernel.  Same images and weights bit for bit on every route. */
int stnerf_composite(const float* t, const float* raw, const uint8_t* mask, int64_t n, int l, int S,
                     const stnerf_composite_params* params_host, float* layer_out, float* mixed_out,
                     float* weights, int32_t* order, uint8_t* scratch, stnerf_stream_t stream);

/* backward of stnerf_composite -- what loss.backward() (engine/layered_trainer.py:277) does to
 * VolumeRenderer.forward / gen_weight (layers/render_layer.py:8-58) and to the merge gather
 * (modeling/layered_rfrender.py:425-429, :587-592) through ATen: given g_layer[n][l][5] and g_mixed[n][5] = dLoss /
 * d{colour(3), depth, acc} of every layer's composite and of the merged one (either may be NULL), writes
 * d_raw[n][l][S][4] = dLoss / d{rgb(3), sigma} of the raw network outputs (zeros where a layer was not evaluated on a ray).
 * t, raw, mask, params: exactly what the forward call got; order[n][l*S]: the forward's `order` output (required with
 * g_mixed).  The density edits act as in the reference's in-place writes: an overwritten sigma gets no gradient, layer 2's
 * `*= alpha` scales it.  Depths get none (the sampler and sample_pdf are detached, :314-315, :460-461).
 * torch.cumprod's backward is reproduced as ATen computes it without zeros in the input (reversed cumsum / input; the
 * 1e-10 of gen_weight keeps every factor positive).  csrc/render_bwd.hip. */
int stnerf_composite_bwd(const float* t, const float* raw, const uint8_t* mask, const int32_t* order, int64_t n, int l, int S,
                         const stnerf_composite_params* params_host, const float* g_layer, const float* g_mixed,
                         float* d_raw, stnerf_stream_t stream);

/* The launch plan stnerf_composite follows for a shape (host arithmetic only, no GPU needed): plan[0] 1 = the LDS-staged
 * kernel alone; [1] single-layer pre-pass (0 none, 1 / 2 = its two instantiations); [2] launches of the merge kernel (1 or
 * 2); [3] layers the first launch's merged list holds; [4] 1 = scratch is cleared first; [5], [6] waves per workgroup and
 * dynamic LDS bytes of the first launch (or of the staged kernel); [7], [8] of the second launch (0 without one).
 * STNERF_EINVAL when a ray of l * S samples does not fit the LDS (the message says how much it needs). */
int stnerf_composite_plan(int l, int S, int with_scratch, int with_order, int64_t* plan);

/* a13 (+ the sort/merge and point generation of layered_rfrender.py:459-475): inverse-CDF
 * resampling of every layer.  utils/sample_pdf.py:18-63.
 * t[n][l][n1] coarse depths, weights[n][l][n1] coarse per-layer weights (interior [1:-1] used),
 * u: [l][n][n2] uniform draws to replay, or NULL -> device Philox (seed, global ray index, stream 1).
 * Outputs: t_fine[n][l][n1+n2] ascending; xyz_fine[n][l][n1+n2][3] (may be NULL) = un-edited points;
 * optional debug/parity outputs z_new[n][l][n2] (unsorted new samples), inds[n][l][n2] int32
 * (searchsorted index), cdf[n][l][n1-1].
 * Exactness: pdf = (w + 1e-5) / torch.sum(w + 1e-5) with the sum in ATen's CPU reduction order (8-float vectors x 4
 * interleaved accumulators), cdf = torch.cumsum accumulated in fp64 and rounded per prefix as ATen's CPU kernel does:
 * cdf, inds and z_new are bit-equal to the reference's CPU evaluation of utils/sample_pdf.py on the same (t, w, u).
 * mask (may be NULL): stnerf_sample_coarse's byte per (ray, layer).  A pair whose "missed" hint (bit 1) is set is SKIPPED:
 * every one of its depths is -1000, so are its resampled depths, and nobody reads them -- stnerf_composite with the same mask
 * never looks at a flagged layer's depths, the networks list hit rays only: its t_fine / xyz_fine rows are left unwritten
 * (16 B x S per pair: at nine layers most of this kernel's bytes were such constant fills).  Without hints (NULL, or bit 1 clear)
 * a pair whose depths are all -1000 gets t_fine = -1000 and the matching points, as before. */
int stnerf_resample(const float* t, const float* weights, int64_t n, int l, int n1, int n2,
                    const float* u, uint64_t seed, int64_t ray_index_base, int64_t ray_index_stripe,
                    int64_t ray_index_period, const float* rays,
                    int ray_stride, const stnerf_layer_edit* edits_host, const float* pivot_host, const uint8_t* mask,
                    float* t_fine, float* xyz_fine, float* z_new, int32_t* inds, float* cdf,
                    stnerf_stream_t stream);

/* ---- backward pass of the two networks (csrc/train.hip) ------------------------------------------------------
 * What engine/layered_trainer.py:192-217's loss.backward() does to modeling/spacenet.py:45-86 / modeling/motion_net.py:20-32
 * through ATen (addmm_backward: two mm + a sum per nn.Linear), as f32 MFMA GEMMs (v_mfma_f32_32x32x2_f32, exact fp32
 * products and accumulation) on matrices the caller owns.  stnerf_amd.modeling.autograd strings them into
 * torch.autograd.Functions for SpaceNet / MotionNet: the forward is recomputed chunk by chunk with every layer's input kept
 * (sample-major [rows][ld]), then walked backwards.  All row strides (ld*) in floats; operand rows that are read with 16-byte
 * vectors need ld % 4 == 0, a 16-byte aligned base and allocations covering whole vectors (pad columns: weights zero).
 *
 * y[m][n] = act(sum_k x[m][k] w[n][k] + bias[n])   (w: the reference's nn.Linear layout, out x in; relu = 0 / 1) */
int stnerf_train_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t m, int n, int k,
                            int relu, float* y, int64_t ldy, stnerf_stream_t stream);
/* dx[m][k] (+)= (sum_n dy[m][n] w[n][k]) * (mask[m][k] > 0): `mask` = the stored post-ReLU tensor that fed this layer (the
 * ReLU backward of the PRODUCING layer folded in; NULL: none); accumulate = 1 adds to dx (a tensor with two consumers). */
int stnerf_train_linear_dx(const float* dy, int64_t lddy, const float* w, int64_t ldw, int64_t m, int n, int k, const float* mask,
                           int64_t ldmask, int accumulate, float* dx, int64_t lddx, stnerf_stream_t stream);
/* dw[n][k] (+)= sum_m dy[m][n] x[m][k];  db[n] (+)= sum_m dy[m][n] (db may be NULL).  The samples are cut into <= 256 slices
 * of >= 256 samples (as many as give the launch two workgroups per CU), reduced by separate workgroups into `workspace`, then
 * summed in slice order: deterministic, no atomics. */
int64_t stnerf_train_dw_workspace_bytes(int64_t m, int n, int k);
int stnerf_train_linear_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t m, int n, int k, float* dw, int64_t lddw,
                           float* db, int accumulate, void* workspace, int64_t workspace_bytes, stnerf_stream_t stream);
/* Every weight and bias gradient of a network in ONE launch (+ one for the reduction): `count` (<= 16) layers whose dy / x have
 * the same m rows -- what loss.backward() accumulates into the .grad of every nn.Linear of modeling/spacenet.py:45-86 or
 * modeling/motion_net.py:20-32 --, same operand rules and the same result contract as stnerf_train_linear_dw (dw[n][k] (+)= dy^T
 * x, db[n] (+)= column sums of dy, db may be NULL).  csrc/train_dw.hip: a WAVE keeps a 128 x 128 tile of one dw in its
 * accumulator registers (row blocks of 128 -- one block of 32 for a layer of <= 4 outputs --, column pieces of 128, the last piece
 * 96 / 64 / 32 wide) and streams a range of samples past it, operands straight from the row-major matrices (no LDS); sample ranges
 * are sized by a tile's MFMA count so that the launch's <= 1024 waves finish together (>= 64 samples each); partial tiles are summed
 * in range order: deterministic, no atomics.  At most 64 tiles and 100 reduction segments (tiles + bias row blocks) per call.
 * Columns of dy / x beyond n / k may be read (up to the tile's width: into the row's padding, the next row, never beyond the last
 * row's round4 end); they only reach outputs that are not stored. */
typedef struct stnerf_dw_problem {
    const float* dy;   /* [m][lddy], n columns */
    int64_t lddy;
    const float* x;    /* [m][ldx], k columns */
    int64_t ldx;
    float* dw;         /* [n][lddw] */
    int64_t lddw;
    float* db;         /* [n] or NULL */
    int32_t n, k;
} stnerf_dw_problem;
int64_t stnerf_train_dw_batch_workspace_bytes(const stnerf_dw_problem* problems, int32_t count, int64_t m);
int stnerf_train_dw_batch(const stnerf_dw_problem* problems, int32_t count, int64_t m, int32_t accumulate, void* workspace,
                          int64_t workspace_bytes, stnerf_stream_t stream);
/* Positional encoding (utils/dimension_kernel.py:54-73) of x[src][0..dim) into columns [col0, col0 + dim (include_input + 2
 * n_freq)) of y[rows][ldy] (a column block of a layer's input matrix: the skip connection of modeling/spacenet.py:128 and
 * rgb_net's input :141-151 are assembled in place).  rows_per_src = ns repeats a ray's encoding on its ns samples (:115,118);
 * relu = 1 applies rgb_net's leading in-place ReLU to the encoded columns (:80); lerp_col >= 0 treats that input column as
 * a frame id and blends the encodings of floor(t) and floor(t) + 1 (modeling/motion_net.py:52-60), -1: none. */
int stnerf_train_encode(const float* x, int64_t ldx, int dim, int n_freq, int include_input, int64_t rows, int rows_per_src, int relu,
                        int lerp_col, float* y, int64_t ldy, int col0, stnerf_stream_t stream);
/* dx[row][j] (+)= sum_features dy[row][col0 + feature] d enc_feature / d x_j for the first dim_out input columns. */
int stnerf_train_encode_bwd(const float* x, int64_t ldx, int dim, int n_freq, int include_input, int64_t rows, const float* dy,
                            int64_t lddy, int col0, int dim_out, int accumulate, float* dx, int64_t lddx, stnerf_stream_t stream);

/* The SpaceNet's backward as fused launches (csrc/train_wave.hip, csrc/mlp_wave.hip; networks without deep_rgb):
 *
 * stnerf_train_spacenet_fwd = stnerf_mlp_stage on ONE SpaceNet (exact f32, the arithmetic of the training forward), every ray
 * evaluated, which ALSO writes each layer's input as it passes through the registers: act[s][row][ld_act[s]], row = ray * ns + k,
 * s = 0 .. 6 the post-ReLU outputs of stage1.0 .. stage2.4 (256 columns), s = 7 of rgb_net.1 (128 columns), and pe[row][ld_pe] =
 * PE_10(pos) (63 columns + one zero) -- the right operands of the weight gradients -- and relu_bits (uint32, may be NULL): stage s's
 * plane starts at relu_bits + s * relu_bits_stride and holds 8 words per row: the ReLU masks [act_s > 0] as bits, 32 bytes per row and
 * layer in the lane order of the wave kernels (csrc/mlp_wave.hip StoreTap) -- all stnerf_train_spacenet_dx reads of the activations
 * (a row range of a larger launch's planes is addressed by offsetting the pointer and keeping the stride).  The matrices are the caller's (16-byte aligned, row
 * strides multiples of 4 floats), e.g. column blocks of wider ones (stage2.0's input [h4 | PE]).
 * act_host / ld_act_host: host arrays of 8 device pointers / strides.  queue, ray_bias: as for stnerf_mlp_stage (one layer).
 *
 * stnerf_train_spacenet_dx: from d_raw[rows][4] = dLoss / d {r, g, b, sigma} walks the layers backwards, a wave carrying the
 * gradient of its 32 rows from layer to layer in registers: d act_{s-1} = (d act_s * [act_s > 0]) W_s (threshold_backward, then
 * addmm_backward's grad_input), the masks from relu_bits.  dy[s][row][ld_dy[s]] receives the masked gradient = dLoss / d
 * (pre-activation of that layer), the left operand of its weight gradient (stnerf_train_linear_dw); dpe (may be NULL) receives
 * dLoss / d PE(pos) (64 columns), for stnerf_train_encode_bwd.  wt: the transposed weights, sections [out / 4][N][4] (N = inputs
 * padded to a multiple of 32) at offsets_host[0] (rgb_net.1's 256 backbone columns: out 128), [1 .. 7] (stage1.0: N 64; stage1.2 ..
 * 1.6: N 256; stage2.0: N 320 = [h4 | PE]; stage2.2, 2.4: N 256), then density_net.0's 256 weights at [8] and the colour head's
 * [3][128] at [9] (float offsets, multiples of 4); stnerf_amd.modeling.autograd builds it. */
int stnerf_train_spacenet_fwd(int kind, const void* packed, int64_t n_rays, int ns, const float* xyz, int64_t xyz_ray_stride,
                              const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride, float* raw,
                              int64_t raw_ray_stride, float* const* act_host, const int32_t* ld_act_host, float* pe, int32_t ld_pe,
                              uint32_t* relu_bits, int64_t relu_bits_stride, uint32_t* queue, float* ray_bias, stnerf_stream_t stream);
/* The same launch in split bf16 (round 6): stnerf_mlp_stage's STNERF_STAGE_BF16X3 kernel (csrc/mlp_bf16x3.hip) with the same tap --
 * raw is bit-identical to what stnerf_mlp_stage / the render pipeline give for these samples in that arithmetic, the stored
 * activations are that kernel's fp32 layer outputs (fp32-faithful: within a few ulp of the exact-f32 launch's), the masks are
 * theirs.  `packed`: a blob of stnerf_pack_net_bf16x3[_device], 1 KB aligned.  Everything else as above. */
int stnerf_train_spacenet_fwd_bf16x3(int kind, const void* packed, int64_t n_rays, int ns, const float* xyz, int64_t xyz_ray_stride,
                                     const float* dirs, int64_t dirs_ray_stride, const float* times, int64_t times_ray_stride, float* raw,
                                     int64_t raw_ray_stride, float* const* act_host, const int32_t* ld_act_host, float* pe, int32_t ld_pe,
                                     uint32_t* relu_bits, int64_t relu_bits_stride, uint32_t* queue, float* ray_bias,
                                     stnerf_stream_t stream);
int stnerf_train_spacenet_dx(const float* wt, const uint32_t* offsets_host, const float* d_raw, int64_t rows, const uint32_t* relu_bits,
                             int64_t relu_bits_stride, float* const* dy_host, const int32_t* ld_dy_host, float* dpe, int32_t ld_dpe,
                             stnerf_stream_t stream);
/* stnerf_train_spacenet_dx in split bf16 (round 6; csrc/mlp_bf16x3.hip: train_space_dx_bx_kernel -- the forward kernel's machinery with
 * the transposed weights as a bf16x3 stream through the LDS ring, the masks fetched per work item by LDS-DMA): same inputs, same dy
 * matrices, fp32-faithful values.  The weights come as ONE blob, stnerf_pack_dx_bf16x3_device(kind, weights_dev (the network's 10
 * weight tensors in reference layout, host array of device pointers), 10, with_dpos, dst (1 KB aligned,
 * stnerf_packed_bytes_dx_bf16x3(kind, with_dpos) bytes)).  with_dpos != 0: the chain goes on to PE(pos) and leaves
 * dLoss / d PE(pos) as the SUM of two 64-column matrices, dpe (d y0 W_stage1.0) and dpe_skip (d y4 W_stage2.0[:, 256:]) -- the caller
 * adds them (the exact-f32 kernel carries one through four layers in registers this kernel does not have); with_dpos == 0: both NULL. */
int64_t stnerf_packed_bytes_dx_bf16x3(int kind, int with_dpos);
int stnerf_pack_dx_bf16x3_device(int kind, const float* const* weights_dev, int n_tensors, int with_dpos, void* dst_dev, int64_t dst_bytes,
                                 stnerf_stream_t stream);
int stnerf_train_spacenet_dx_bf16x3(const void* packed_dx, int with_dpos, const float* d_raw, int64_t rows, const uint32_t* relu_bits,
                                    int64_t relu_bits_stride, float* const* dy_host, const int32_t* ld_dy_host, float* dpe, int32_t ld_dpe,
                                    float* dpe_skip, int32_t ld_dpe_skip, stnerf_stream_t stream);

/* The MotionNet's backward (modeling/motion_net.py:20-71 under loss.backward()) as the same two launches (csrc/train_wave.hip):
 *
 * stnerf_train_motionnet_fwd: flow[row][3] = MotionNet(xt[row] = {x, y, z, frame id}) with the arithmetic of the inference kernels'
 * MotionNet (exact f32; motion_flags: STNERF_MOTION_PLAIN_TIME or 0, as stnerf_motionnet_fwd), writing what the backward needs as
 * the rows pass through the registers: enc[row][ld_enc] = the 84 encoded features (+ 4 zeros; fractional frame ids blended as
 * motion_net.py:52-60 does), act[s][row][ld_act[s]] = the post-ReLU outputs of motion_net.0, .2, .4, .6, .8 (128 columns), and
 * relu_bits: stage s's plane at relu_bits + s * relu_bits_stride, 4 words per row (the masks [act_s > 0], 16 bytes per row and layer,
 * in the lane order of the wave kernels).
 *
 * stnerf_train_motionnet_dx: from d_flow[rows][4] (dLoss / d flow in columns 0 .. 2) back through the five hidden layers with the
 * gradient in registers: dy[s][row][ld_dy[s]] = dLoss / d (pre-activation of motion_net.{0,2,4,6,8}[s]) (128 columns), the left
 * operands of the weight gradients, and denc[row][ld_denc] (may be NULL; 96 columns written, 84 meaningful) = dLoss / d encoding
 * for stnerf_train_encode_bwd.  wt: transposed sections [128 / 4][128][4] of motion_net.0 (its 84 inputs zero-padded to 128) and
 * .2 .. .8 at offsets_host[0 .. 4], the flow head's [3][128] as it is at [5] (float offsets, multiples of 4). */
int stnerf_train_motionnet_fwd(const void* packed, const float* xt, int64_t rows, int motion_flags, float* flow, float* enc, int32_t ld_enc,
                               float* const* act_host, const int32_t* ld_act_host, uint32_t* relu_bits, int64_t relu_bits_stride,
                               stnerf_stream_t stream);
int stnerf_train_motionnet_dx(const float* wt, const uint32_t* offsets_host, const float* d_flow, int64_t rows, const uint32_t* relu_bits,
                              int64_t relu_bits_stride, float* const* dy_host, const int32_t* ld_dy_host, float* denc, int32_t ld_denc,
                              stnerf_stream_t stream);

/* The whole chunk pipeline of LayeredRFRender.forward (modeling/layered_rfrender.py:141-734) behind one call:
 * coarse sampler -> mask compaction -> [MotionNet] -> SpaceNets -> composite/merge -> resample -> [MotionNet] ->
 * fine SpaceNets -> composite/merge, all enqueued on `stream` into a caller-provided workspace.  Host-side
 * box interpolation / editing (l x 8 x 3 numbers, :190-242) stays with the caller, who passes the edited boxes
 * and the inverse point edits.  Packed-network pointers are device blobs of stnerf_pack_net[_bf16x3]. */
typedef struct stnerf_nets {
    const void* bkgd;                           /* bkgd_spacenet            (STNERF_NET_SPACE)               */
    const void* bkgd_fine;                      /* bkgd_spacenet_fine                                        */
    const void* space[STNERF_MAX_LAYERS];       /* [i] = spacenets[i-1], i >= 1 ([0] unused)                 */
    const void* space_fine[STNERF_MAX_LAYERS];  /* [i] = spacenets_fine[i-1]                                 */
    const void* motion[STNERF_MAX_LAYERS];      /* [i] = time_deform_nets[i-1] (use_deform_time only);
                                                   [0] = bkgd_time_deform_net (bkgd_use_deform_time only)    */
} stnerf_nets;

typedef struct stnerf_render_params {
    int32_t l, n1, n2;            /* layers incl. background, coarse / fine sample counts                     */
    int32_t ray_stride;           /* floats per ray: 6 + l (retiming) or 7                                   */
    int32_t retiming;             /* 1: frame id of layer i in column 6+i; 0: per-ray frame id in column 6   */
    int32_t only_coarse;
    int32_t use_deform_time, use_space_time;
    int32_t precision;            /* 3: bf16x3 (split-bf16, stnerf_mlp_stage with STNERF_STAGE_BF16X3; nets packed by
                                     stnerf_pack_net_bf16x3) -- what the Python boundary selects by default; 0: exact f32 MFMA, one
                                     persistent stnerf_mlp_stage launch per stage; 2: exact f32, one launch per network
                                     (round-1 scheduling, for A/B runs).  (1 was the retired split-fp16 mode: rejected.) */
    int32_t has_edits;            /* edits_* / pivot are meaningful                                          */
    int32_t bkgd_use_deform_time; /* BKGD_USE_DEFORM_TIME: nets.motion[0] warps the background samples (:358-367) */
    int32_t bkgd_use_space_time;  /* BKGD_USE_SPACE_TIME: background SpaceNets take the frame id (needs use_space_time, :382-390) */
    int32_t deep_rgb;             /* DEEP_RGB && USE_SPACE_TIME (:35): every SpaceNet blob is of a *_DEEP kind */
    int32_t shown[STNERF_MAX_LAYERS];                 /* display_layers (:99-112); [0] ignored               */
    float border, near, alpha;                        /* BOARDER_WEIGHT, model.near, model.alpha             */
    float density_threshold, bkgd_density_threshold;  /* applied in retiming mode only, as the reference     */
    uint64_t seed;                                    /* device RNG (used where jitter / u are NULL)         */
    int64_t ray_index_base, ray_index_stripe, ray_index_period;  /* ray window of rays[0..n) (see above)     */
    stnerf_layer_edit edits_coarse[STNERF_MAX_LAYERS]; /* :293-303 */
    stnerf_layer_edit edits_fine[STNERF_MAX_LAYERS];   /* :467-475 */
    float pivot[3];
} stnerf_render_params;

int64_t stnerf_render_workspace_bytes(int64_t n, int l, int n1, int n2, int only_coarse);
/* Outputs: mixed_*[n][5], layer_*[n][l][5] = {colour(3), depth, acc}; mask[n][l].  jitter [l][n][n1] / u [l][n][n2]
 * replay uniform draws (NULL = device RNG).  With only_coarse the fine outputs may be NULL. */
int stnerf_render_rays(const float* rays, int64_t n, const float* boxes, int64_t box_ray_stride,
                       const stnerf_nets* nets_host, const stnerf_render_params* params_host, const float* jitter,
                       const float* u, void* workspace, int64_t workspace_bytes, float* mixed_fine,
                       float* mixed_coarse, float* layer_fine, float* layer_coarse, uint8_t* mask,
                       stnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STNERF_H */
