"""The oracle's training step against the reference's: loss, every parameter's gradient and the parameters
after one Adam step, recorded by tests/golden/make_golden.py --grads from the reference's OWN nn.Modules, loss and optimiser.
This pins the oracle's autograd (what the GPU tests compare the HIP backward with in fp64) to the reference's."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import stnerf_oracle as O                    # noqa: E402
from stnerf_amd import synthetic as syn                  # noqa: E402
from train_step_common import compare_digest, load_fixture, oracle_step   # noqa: E402


@pytest.mark.parametrize("name", ["train_c3", "train_coarse_only", "train_c4", "train_flags", "train_same_spacenet", "train_bkgd_time", "train_tf_c3"])
def test_oracle_training_step_matches_the_reference(name):
    z, meta = load_fixture(name)
    sd, out, loss, parts = oracle_step(z, meta, torch.float32)
    assert float(loss) == pytest.approx(float(z["loss"][0]), rel=1e-6)
    for k, v in parts.items():
        assert float(v) == pytest.approx(float(z[k][0]), rel=1e-5, abs=1e-9), k
    assert torch.allclose(out[0][0], torch.from_numpy(z["fine_mixed_color"]), atol=1e-6)
    recorded = [k.split("|", 1)[1] for k in z.files if k.startswith("grad|")]
    assert len(recorded) >= 20 and not meta["without_grad"] or name == "train_coarse_only"
    worst = 0.0
    for pname in recorded:
        g = sd[pname].grad
        assert g is not None, pname
        worst = max(worst, compare_digest(pname, syn.tensor_digest(pname, g, meta["grad_samples"]), z["grad|" + pname], rel=2e-5))
    assert worst <= 1.0, worst
    # parameters the reference left without a gradient (the fine networks of a coarse-only epoch) get none here either
    for pname in meta["without_grad"]:
        assert sd[pname].grad is None or float(sd[pname].grad.abs().max()) == 0.0, pname
    # one Adam step (solver/build.py:18: betas (0.9, 0.999), no weight decay)
    params = [sd[p] for p in recorded]
    opt = torch.optim.Adam(params, lr=meta["lr"], betas=(0.9, 0.999), weight_decay=0.0)
    opt.step()
    # (Adam's first step moves an entry by lr g / (|g| + 1e-8): where |g| is of the order of 1e-8 .. 1e-6 the step follows the
    # gradient's last bits -- train_c4 has one such tensor, the flow head of a performer few rays hit: at most two tensors may
    # exceed the 2e-6 bar, by less than 5 x)
    ratios = {pname: compare_digest(pname, syn.tensor_digest(pname, sd[pname], meta["grad_samples"]), z["stepped|" + pname], rel=2e-6)
              for pname in recorded}
    over = {k: v for k, v in ratios.items() if v > 1.0}
    assert len(over) <= 2 and all(v <= 5.0 for v in over.values()), over


def test_teacher_forced_oracle_step_sits_on_the_reference_fixture():
    """The oracle fed the reference's own fine depths and deformed points (train_tf_c3): in fp32 the same gradients as free-running (the
    oracle's fp32 chain already reproduces the reference's positions bit for bit on the CPU); in fp64 -- what the GPU test judges
    ReLU'(0) events and fp32 summation noise by -- every gradient within 2e-5 of the reference's: with the positions forced the
    conditioning of sin(2^9 x) behind the resampler and the deformation nets is out of the comparison."""
    z, meta = load_fixture("train_tf_c3")
    for dtype, bar in ((torch.float32, 2e-5), (torch.float64, 2e-5)):
        sd, _, loss, _ = oracle_step(z, meta, dtype, sample_dtype=torch.float32, teacher=True)
        assert float(loss) == pytest.approx(float(z["loss"][0]), rel=1e-6)
        errs = {k: compare_digest(k, syn.tensor_digest(k.split("|", 1)[1], sd[k.split("|", 1)[1]].grad.float(), meta["grad_samples"]), z[k], rel=bar)
                for k in z.files if k.startswith("grad|")}
        assert max(errs.values()) <= 1.0, (dtype, max(errs, key=errs.get), max(errs.values()))


def test_reference_fp32_gradients_against_an_fp64_evaluation_of_the_same_graph():
    """What the reference's own fp32 autograd is worth as a yardstick.  The fp64 evaluation (on the fp32 run's sample positions)
    agrees with it to fp32 rounding on train_coarse_only -- and on train_c3 for most tensors, but ONE hidden unit of
    bkgd_spacenet.stage2.0 on ONE sample has a pre-activation of 4e-9 in fp64 and <= 0 in ATen's fp32 sgemm: ReLU'(0) passes the
    cotangent in one evaluation and not in the other, which moves that network's gradients by up to 3.7 % of a tensor's largest
    entry.  The GPU test therefore accepts either side of such an event (tests/test_gpu_training.py)."""
    z, meta = load_fixture("train_coarse_only")
    sd, _, loss, _ = oracle_step(z, meta, torch.float64, sample_dtype=torch.float32)
    assert float(loss) == pytest.approx(float(z["loss"][0]), rel=1e-6)
    errs = [compare_digest(k, syn.tensor_digest(k.split("|", 1)[1], sd[k.split("|", 1)[1]].grad, meta["grad_samples"]), z[k], rel=2e-5)
            for k in z.files if k.startswith("grad|")]
    assert max(errs) <= 1.0, max(errs)
    z, meta = load_fixture("train_c3")
    sd, _, loss, _ = oracle_step(z, meta, torch.float64, sample_dtype=torch.float32)
    assert float(loss) == pytest.approx(float(z["loss"][0]), rel=1e-6)
    errs = {k: compare_digest(k, syn.tensor_digest(k.split("|", 1)[1], sd[k.split("|", 1)[1]].grad, meta["grad_samples"]), z[k], rel=1.0)
            for k in z.files if k.startswith("grad|")}
    ranked = sorted(errs.values())
    assert ranked[len(ranked) // 2] <= 2e-5 and ranked[-1] <= 0.05, (ranked[len(ranked) // 2], ranked[-1])
    assert max(errs, key=errs.get).startswith("grad|bkgd_spacenet.")
