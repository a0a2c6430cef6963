"""Build-quality gate for the wave stage kernel (csrc/mlp_wave.hip), CPU only: hipcc cross-compiles it to gfx950 assembly.

The kernel runs one wave per SIMD with the whole unified register file; a change that pushes a handful of values into
scratch does not fail any numerical test, it just slows every phase down (scratch reloads queue up in front of the
operand loads of the K loop: -3 % measured).  So the resource usage is asserted here, not looked at by hand.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_wave_stage_kernel_uses_no_scratch_and_one_wave_per_simd(tmp_path):
    src = os.path.join(ROOT, "st-nerf_amd", "csrc", "mlp_wave.hip")
    asm = tmp_path / "mlp_wave.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", str(asm), src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=600)
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN6stnerf21mlp_wave_stage_kernelILb[01]E\S*):", text, re.M)
    # the plain and the deep_rgb variant, and (round 5) the training instantiation of the plain one that writes every layer's input out
    assert len(kernels) == 3 and sum("StoreTapArgs" in k for k in kernels) == 1, kernels
    scratch = [int(v) for v in re.findall(r"; ScratchSize: (\d+)", text)]
    vgprs = [int(v) for v in re.findall(r"; TotalNumVgprs: (\d+)", text)]
    occupancy = [int(v) for v in re.findall(r"; Occupancy: (\d+)", text)]
    assert scratch == [0, 0, 0], f"the wave stage kernel spills to scratch: {scratch} bytes"
    assert all(v <= 512 for v in vgprs) and len(vgprs) == 3, vgprs
    assert occupancy == [1, 1, 1], occupancy
    # the hot loop is what it is supposed to be: f32 MFMAs fed from registers, accumulators loaded by LDS reads directly
    assert text.count("v_mfma_f32_32x32x2_f32") >= 3 * 2400
    assert "scratch_load" not in text and "scratch_store" not in text


def _compile(src_name, tmp_path, extra=()):
    src = os.path.join(ROOT, "st-nerf_amd", "csrc", src_name)
    asm = tmp_path / (src_name + ".s")
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", *extra,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", str(asm), src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=600)
    return str(asm)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_wave_stage_kernel_vector_instruction_ceiling(tmp_path):
    """The exact-f32 kernel's distance from the MFMA pipe's own rate IS its vector-instruction count (every VALU /
    v_accvgpr instruction takes 5 - 6 cycles out of the f32 MFMA stream): a ceiling per region of
    the work-item loop, from tools/isa_vector_count.py's static count (measured: 1027 / 129 / 779 / 265 / 926), so that a
    change of compiler or source that adds vector work shows up here and not as a lost percent on the GPU."""
    out = subprocess.run(["python3", os.path.join(ROOT, "tools", "isa_vector_count.py"), "--asm", _compile("mlp_wave.hip", tmp_path)],
                         check=True, capture_output=True, text=True, timeout=600).stdout
    got = {m.group(1).strip(): int(m.group(2)) for m in re.finditer(r"^(.+?)\s+vec\s+(\d+)\s", out, re.M)}
    ceilings = {"item head .. MotionNet loop": 1080, "MotionNet layer loop body (x 4)": 136, "MotionNet tail + encoding + stage1.0": 820,
                "SpaceNet layer loop body (x 6)": 272, "sigma + rgb_net.1 + head + store": 975}
    assert set(ceilings) <= set(got), out
    for region, cap in ceilings.items():
        assert got[region] <= cap, f"{region}: {got[region]} vector instructions (ceiling {cap})\n{out}"
    total = int(re.search(r"background path .*?: (\d+) vector", out).group(1))
    assert total <= 4500, out


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_bf16x3_stage_kernel_resources(tmp_path):
    """csrc/mlp_bf16x3.hip: one wave per SIMD; (almost) no scratch traffic between the first and the last MFMA of a work
    item -- a reload there waits behind vmcnt(0), i.e. behind the weight ring's DMA queue, inside the K passes.  Tolerated:
    a few loop-invariant values reloaded in front of the first MFMA (the 1 / ns reciprocal of the row lookup ...), ONE
    reload in front of rgb_net.1's last slot (the ray index of the C-operand row, which waits for memory there anyway) and
    the output address behind the last MFMA.  The LDS-DMA / barrier / MFMA structure is what the design says: 48 MFMAs and
    24 operand reads per ring slot."""
    text = open(_compile("mlp_bf16x3.hip", tmp_path)).read()
    kernels = re.findall(r"^(_ZN6stnerf23mlp_bf16x3_stage_kernelILb[01]E\S*):", text, re.M)
    # the two inference kernels (deep_rgb or not) and the training tap's variant (round 6)
    assert len(kernels) == 3 and sum("StoreTapArgs" in k for k in kernels) == 1, kernels
    for name in kernels:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = body.split("\n")
        loop = next(i for i, l in enumerate(lines) if "This Loop Header: Depth=1" in l)      # the work-item loop
        in_loop = "\n".join(lines[loop:])
        mf = [i for i in range(loop, len(lines)) if "v_mfma" in lines[i]]
        inside = [l for l in lines[mf[0]:mf[-1]] if "scratch_" in l]
        tap = "StoreTapArgs" in name
        assert len(inside) <= (0 if tap else 1), inside[:5]
        assert sum("scratch_" in l for l in lines[loop:mf[0]]) <= 8 and sum("scratch_" in l for l in lines[mf[-1]:]) <= 6
        assert "s_swappc" not in body          # (the tap variant's lambdas are forced inline: a real call spills the wave)
        n_mfma, n_read = in_loop.count("v_mfma_f32_32x32x16_bf16"), len(re.findall(r"ds_read_b128 a\[", in_loop))
        n_bar, n_dma = in_loop.count("s_barrier"), in_loop.count("global_load_lds_dwordx4")
        deep = "ILb1E" in name
        # ring slots as LISTED (loop bodies once): MotionNet 3 + 4 (compiled out of the tap variant), stage1.0 2 x 2, three copies of a
        # 256-wide layer (2 x 8), stage2.0's PE slots 2 x 2, rgb_net.1 7 + 1, the deep_rgb loop body 4
        slots = (0 if tap else 3 + 4) + 4 + 3 * 16 + 4 + 8 + (4 if deep else 0)
        assert n_mfma == 48 * slots, (name, n_mfma, 48 * slots)
        assert 24 * slots <= n_read <= 24 * slots + 16 * 12, (name, n_read)    # + the C-operand (bias) reads: 16 per pass start
        assert n_bar >= slots and n_dma >= 6 * slots, (name, n_bar, n_dma)
        if tap:      # as listed: (stage1.0 2 + three layer copies 6 + rgb_net.1 1) boundary passes x 16 + PE(pos) 8 activation stores + the
                     # output, all global (a buffer store's descriptor does not fit at a boundary: see the tap's comment)
            assert in_loop.count("global_store_dwordx4") == 9 * 16 + 8 + 1 and "buffer_store" not in in_loop, in_loop.count("global_store_dwordx4")
            assert "s_waitcnt vmcnt(24)" in in_loop      # the slot turns behind a boundary count its 18 stores in
    # ---- the backward chain (train_space_dx_bx_kernel<DPOS>): no scratch, no call, the ring structure, the masks' reads as asm
    dx = re.findall(r"^(_ZN6stnerf24train_space_dx_bx_kernelILb[01]E\S*):", text, re.M)
    assert len(dx) == 2, dx
    for name in dx:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body and "s_swappc" not in body, name
        dpos = "ILb1E" in name
        # rgb_net.1 2 x 4, the two loop bodies and stage2.0's layer 16 each; with d pos: two half passes of 4
        slots = 8 + 3 * 16 + (8 if dpos else 0)
        assert body.count("v_mfma_f32_32x32x16_bf16") == 48 * slots, (name, body.count("v_mfma_f32_32x32x16_bf16"))
        # what drains the weight ring's DMA queue: the item's mask fetch (top of the item) and the kernel's end -- not the boundaries
        assert body.count("s_waitcnt vmcnt(0)") <= 3, body.count("s_waitcnt vmcnt(0)")
    for name in kernels + dx:
        tail = text[text.index(name + ":"):]
        occ = int(re.search(r"; Occupancy: (\d+)", tail).group(1))
        scr = int(re.search(r"; ScratchSize: (\d+)", tail).group(1))
        assert occ == 1 and scr <= (0 if ("StoreTapArgs" in name or "dx_bx" in name) else 128), (name, occ, scr)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_compositor_and_resampler_production_kernels_resources(tmp_path):
    """csrc/render.hip: the compositor's merge kernel and the resampler are bound by vector-instruction issue and by how many
    waves hide their LDS / HBM round trips (DESIGN.md sections 4.2 / 4.3), so the flavours the BASELINE shapes run (S a multiple
    of 64; device draws) must keep their occupancy targets without scratch, use the one-instruction in-place DPP scan steps,
    and the resampler's sort must not go through ds_bpermute except for its single distance-32 stage."""
    text = open(_compile("render.hip", tmp_path)).read()

    def kernel(mangled_part):
        name = next(n for n in re.findall(r"^(_ZN6stnerf\w+):", text, re.M) if mangled_part in n)
        body = text[text.index(name + ":"):]
        end = body.index("s_endpgm")
        meta = body[end:end + 6000]
        get = lambda k: int(re.search(r"; " + k + r": (\d+)", meta).group(1))
        return body[:end], get("NumVgprs"), get("ScratchSize"), get("Occupancy")

    for part, occupancy in (("composite_merge_kernelILi1ELb1", 7), ("composite_merge_kernelILi2ELb1", 7), ("composite_merge_kernelILi3ELb1", 6),
                            ("resample_kernelILi1ELb1ELb1", 8), ("resample_kernelILi2ELb1ELb1", 7), ("resample_kernelILi2ELb1ELb0", 7)):
        body, vgprs, scratch, occ = kernel(part)
        assert occ >= occupancy, (part, vgprs, occ)
        # (a few loop-invariant scalars parked in scratch outside the hot loops are tolerated for the 7-wave compositor flavours)
        assert scratch <= (48 if "composite" in part else 0), (part, scratch)
        assert "v_mul_f32_dpp" in body or "resample" in part, part
        if "composite" in part:
            assert body.count("v_mul_f32_dpp") >= 6 and body.count("v_add_f32_dpp") >= 30, part      # in-place scans
            assert "ds_or_b32" in body and "v_mbcnt_hi_u32_b32" in body, part                         # slot mask + prefix count
            # the insertion's search: a loop the SCALAR unit controls (interval length in an SGPR), whose body is, per sample
            # block, an add, a ds_read_b32, a compare and a select -- nothing else on the vector unit
            blocks = int(re.search(r"ILi(\d)E", part).group(1))
            loops = [m.group(1) for m in re.finditer(r"Inner Loop Header: Depth=\d+\n((?:.*\n)*?)\s+s_cbranch_scc[01] ", body)]
            search = [b for b in loops if b.count("ds_read_b32") == blocks and "s_lshr_b32" in b]
            assert len(search) == 1, (part, [b.count("ds_read_b32") for b in loops])
            vector = [l.split()[0] for l in search[0].splitlines() if l.strip().startswith("v_")]
            assert len(vector) == 3 * blocks and sum(v.startswith("v_cmp") for v in vector) == blocks \
                and sum(v.startswith("v_cndmask") for v in vector) == blocks, (part, vector)
        else:
            assert body.count("v_min_f32_dpp") >= 14 and body.count("ds_bpermute_b32") <= 2, part    # DPP bitonic stages
            if part.endswith("ELb1ELb1"):   # sizes known at compile time: (almost) no scalar registers parked in vector lanes
                loop = body[body.index("This Loop Header: Depth=1"):]
                parked = len(re.findall(r"v_(?:read|write)lane_b32", loop))
                assert parked <= 80, (part, parked)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_no_valu_exec_write_within_five_wait_states_of_a_dpp_instruction(tmp_path):
    """csrc/render.hip carries hand-written DPP blocks (the in-place wave scans, the bitonic stages).  LLVM's hazard
    recognizer does not look inside inline asm: a VALU write of EXEC (v_cmpx*) needs 5 wait states before a DPP
    instruction reads under it, and a block placed directly behind predicated code would get none from the compiler.  The
    scans pad themselves (s_nop 4 in front of their first step); everything else is checked here on the generated ISA:
    walking back from every *_dpp instruction, fewer than 5 wait states (an instruction = 1, s_nop N = N + 1) may not
    separate it from a v_cmpx.  Branch targets end the walk (another path's padding is that path's business -- and LLVM's
    own hazard pass covers code it generated)."""
    text = open(_compile("render.hip", tmp_path, extra=("-ffp-contract=off",))).read()
    lines = [ln.split(";")[0].strip() for ln in text.split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith(".")]
    n_dpp, offenders = 0, []
    for i, ln in enumerate(lines):
        op = ln.split()[0]
        if not (op.endswith("_dpp") or " row_shr:" in ln or " quad_perm:" in ln or " row_bcast:" in ln or " row_ror:" in ln or " row_shl:" in ln):
            continue
        n_dpp += 1
        waits, j = 0, i - 1
        while j >= 0 and waits < 5:
            prev = lines[j]
            if prev.endswith(":"):                      # a label: stop at the basic-block boundary
                break
            pop = prev.split()[0]
            if pop.startswith("v_cmpx"):
                offenders.append((i, prev, ln, waits))
                break
            waits += (int(prev.split()[1], 0) + 1) if pop == "s_nop" else 1
            j -= 1
    assert n_dpp > 100, n_dpp                            # the scans and sorts are really there
    assert not offenders, offenders[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_fused_backward_kernel_resources(tmp_path):
    """csrc/train_wave.hip (round 5): the dX chain keeps a wave's gradient in registers from the heads to stage1.2 -- one wave per
    SIMD, no scratch, exactly the MFMAs of the chain (16 + 6 x 32 K steps of 32, + the two 2-block products of the PE columns when
    the points need a gradient); the ReLU masks arrive as bit planes (8 loads of 16 bytes per item, one v_bfe_i32 + one v_and per
    value: no activation is read back), every gradient write is a 16-byte vector store."""
    text = open(_compile("train_wave.hip", tmp_path)).read()
    kernels = re.findall(r"^(_ZN6stnerf21train_space_dx_kernelILb[01]EEE\S*):", text, re.M)
    assert len(kernels) == 2, kernels
    # (five kernels in the file: the MotionNet's training forward, the two SpaceNet chains, the two MotionNet chains)
    assert [int(v) for v in re.findall(r"; ScratchSize: (\d+)", text)] == [0] * 5
    assert all(int(v) <= 512 for v in re.findall(r"; TotalNumVgprs: (\d+)", text))
    assert "scratch_" not in text
    motion = re.findall(r"^(_ZN6stnerf22train_motion_dx_kernelILb[01]EEE\S*):", text, re.M)
    assert len(motion) == 2, motion
    for k in motion:
        # four 128 x 128 products (16 K steps x 16 MFMAs), + the one into the encoding whose fourth block nobody stores; five masks of
        # 8 bytes per lane; 5 x 16 gradient stores (+ 12 for d enc)
        body = text[text.index(k + ":"):]
        body = body[:body.index("s_endpgm")]
        dx = "Lb1" in k
        assert body.count("v_mfma_f32_32x32x2_f32") == 4 * 256 + (192 if dx else 0), (k, body.count("v_mfma_f32_32x32x2_f32"))
        assert body.count("global_load_dwordx2") == 5 and body.count("global_load_dwordx4") == 0
        assert body.count("global_store_dwordx4") == 5 * 16 + (12 if dx else 0)
        assert body.count("v_bfe_i32") == 5 * 64 and body.count("v_cmp_") < 16
    fwd = text[text.index("_ZN6stnerf23train_motion_fwd_kernel"):]
    fwd = fwd[fwd.index(":"):fwd.index("s_endpgm")]
    # the tap: 22 quads of the encoding + 16 stores and 2 x 64 (min, shift-or) pairs per layer (rolled layer loop + the last layer)
    assert fwd.count("global_store_dwordx4") >= 11 + 2 * 16 and fwd.count("global_store_dwordx2") == 2 and fwd.count("v_lshl_or_b32") >= 128
    for k in kernels:
        body = text[text.index(k + ":"):]
        body = body[:body.index("s_endpgm")]
        want = (16 + 6 * 32) * 32 + (2 * 32 * 8 if "Lb1" in k else 0)
        assert body.count("v_mfma_f32_32x32x2_f32") == want, (k, body.count("v_mfma_f32_32x32x2_f32"), want)
        assert body.count("global_load_dwordx4") <= 16 and body.count("global_store_dwordx4") >= 7 * 32 + 16
        assert body.count("v_bfe_i32") == (4 + 7 * 8) * 16 and body.count("v_cmp_") < 16


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_weight_gradient_kernel_loops_are_what_the_design_says(tmp_path):
    """csrc/train_dw.hip (round 6): one wave per SIMD, no scratch, and the three inner loops as DESIGN 4.4 describes them --
    the direct f32 tile: 128 MFMAs + 16 buffer loads per 16 samples and NO vector instruction besides (on gfx950 every VALU instruction
    takes its cycles out of the f32 MFMA stream); the f32 bundle: 128 MFMAs, 16 LDS reads, 8 LDS-DMA instructions, one barrier per block;
    the split-bf16 bundle: 192 bf16 MFMAs per two 16-sample steps beside <= 5 vector instructions per MFMA (the two operands' splits),
    32 LDS reads, 16 LDS-DMA instructions, two barriers."""
    from collections import Counter
    text = open(_compile("train_dw.hip", tmp_path)).read()
    assert [int(v) for v in re.findall(r"; ScratchSize: (\d+)", text)] == [0, 0]          # the tile kernel and the reduction
    assert "scratch_" not in text
    assert max(int(v) for v in re.findall(r"; TotalNumVgprs: (\d+)", text)) <= 512
    assert int(re.findall(r"; Occupancy: (\d+)", text)[0]) == 1
    lines = text.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = [x.split()[0] for x in lines[labels[m.group(1)]:i + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            loops.append(Counter(body))
    valu = lambda c: sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
    # innermost loops only (a loop that contains another shows up with its sum: take the smallest body per signature)
    direct = [c for c in loops if c.get("v_mfma_f32_32x32x2_f32") == 128 and c.get("buffer_load_dwordx4") == 16 and not c.get("ds_read_b128")]
    assert direct and any(valu(c) == 0 for c in direct), [dict(c) for c in direct]          # (the tile without bias sums)
    assert all(valu(c) in (0, 32) and c.get("s_waitcnt") == 8 for c in direct), [dict(c) for c in direct]   # 8 waits of vmcnt(14): loads stay 8 steps ahead
    bundle = [c for c in loops if c.get("v_mfma_f32_32x32x2_f32") == 128 and c.get("ds_read_b128") == 16]
    assert bundle and all(c.get("buffer_load_dwordx4") == 8 and c.get("s_barrier") == 1 and valu(c) <= 40 for c in bundle), [dict(c) for c in bundle]
    bx = [c for c in loops if c.get("v_mfma_f32_32x32x16_bf16") == 192 and c.get("ds_read_b128") == 32]   # (innermost: an enclosing loop has more)
    assert len(bx) == 2, [dict(c) for c in bx]                                              # with and without bias sums
    for c in bx:
        assert c.get("ds_read_b128") == 32 and c.get("buffer_load_dwordx4") == 16 and c.get("s_barrier") == 2, dict(c)
        assert c.get("v_cvt_pk_bf16_f32") == 192 and valu(c) <= 5 * 192, (valu(c), dict(c))
