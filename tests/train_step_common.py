"""One iteration of the reference trainer's inner loop (engine/layered_trainer.py:186-283), restated for the tests that compare
a training step with the fixtures tests/golden/make_golden.py --grads recorded from the reference's own model, loss and optimiser
(train_c3.npz, train_coarse_only.npz, train_c4.npz, train_flags.npz)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def replay_of(z, meta, dtype=torch.float32, device="cpu"):
    """The recorded torch.rand draws in the reference's order: l jitter tensors (n, N1), then l resampling tensors (n, N2).  A fixture
    made with seeded draws (make_golden.SeededDraws: the trainer-sized one) holds their seeds and shapes instead: call k =
    torch.rand(shape, generator=manual_seed(base + k)) on the CPU.  A teacher-forced fixture (meta["teacher"]) adds what the reference's
    networks were evaluated on: "z" (l, n, N2) every layer's new fine depths, "xyz_c" / "xyz_f" per performer the deformed points of
    its hit rays (None for layer 0 and for performers no ray hits)."""
    l = meta["L"] + 1
    if "draw_seed_base" in meta:
        draws = [torch.rand(tuple(shape), generator=torch.Generator().manual_seed(meta["draw_seed_base"] + k)).to(dtype)
                 for k, shape in enumerate(meta["draw_shapes"])]
    else:
        draws = [torch.from_numpy(z[f"draw{i}"]).to(dtype) for i in range(meta["n_draws"])]
    rp = {"jitter": torch.stack(draws[:l], 0).to(device)}
    if len(draws) > l:
        rp["u"] = torch.stack(draws[l:2 * l], 0).to(device)
    if meta.get("teacher"):
        rp["z"] = torch.stack([torch.from_numpy(z[f"tf_z{i}"]) for i in range(l)], 0).to(dtype).to(device)
        for key in ("xyz_c", "xyz_f"):
            rp[key] = [None] + [torch.from_numpy(z[f"tf_{key}{i}"]).to(dtype).to(device) if f"tf_{key}{i}" in z.files else None for i in range(1, l)]
    return draws, rp


def trainer_loss(out, rgbs, labels, only_coarse, remove_outliers=True, epoch=1, scalar=100000, penalty=1):
    """engine/layered_trainer.py:211-277: MSE of the coarse and fine mixed colours (layers/loss.py:4) plus, in the first epochs, the
    outlier / inlier losses on every layer's accumulation map."""
    stage2, stage1, stage2_layer, stage1_layer, _ = out
    mse = torch.nn.MSELoss()
    loss1, loss2 = mse(stage1[0], rgbs), mse(stage2[0], rgbs)
    zero = torch.zeros(1, dtype=loss1.dtype, device=loss1.device)
    lm0, lm1 = zero, zero
    if epoch < 3 and remove_outliers:
        o1, o2, i1, i2 = [], [], [], []
        for i in range(len(stage1_layer)):
            if i != 0:
                o1.append(stage1_layer[i][2][labels == 0])
                o2.append(stage2_layer[i][2][labels == 0])
            i1.append(stage1_layer[i][2][labels == i])
            i2.append(stage2_layer[i][2][labels == i])
        i1, i2 = torch.cat(i1, 0), torch.cat(i2, 0)
        lm0, lm1 = torch.sum(torch.abs(1 - i1)), torch.sum(torch.abs(1 - i2))
        if o1:
            lm0 = torch.sum(torch.abs(torch.cat(o1, 0))) * penalty + lm0
            lm1 = torch.sum(torch.abs(torch.cat(o2, 0))) * penalty + lm1
        n = rgbs.shape[0]
        lm0 = lm0 / scalar if lm0 > n * 0.0005 else zero
        lm1 = lm1 / scalar if lm1 > n * 0.0005 else zero
    loss = loss1 + lm0 if only_coarse else loss1 + loss2 + lm0 + lm1
    return loss, dict(loss1=loss1, loss2=loss2, loss_mask_0=lm0, loss_mask_1=lm1)


def compare_digest(name, got, want, rel=2e-5):
    """|got - want| <= rel * max|want| for a stnerf_amd.synthetic.tensor_digest vector (sums are of many entries: judged against
    their own scale).  -> worst ratio error / bound."""
    got, want = got.double(), torch.as_tensor(want).double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = float(want.abs().max())
    if scale == 0.0:
        assert float(got.abs().max()) == 0.0, name
        return 0.0
    err = float((got - want).abs().max())
    return err / (rel * scale)


def oracle_step(z, meta, dtype, sample_dtype=None, teacher=False):
    """The same iteration through the CPU oracle (TEST INFRASTRUCTURE: oracle/stnerf_oracle.py, pinned to the reference's own
    gradients by tests/test_train_step_cpu.py) in ``dtype``; -> (params with .grad, outputs, loss, loss parts).  With
    ``sample_dtype=torch.float32`` an fp64 evaluation sits on the fp32 run's sample positions; with ``teacher`` (a fixture of
    make_golden.py --grads --teacher) also on the reference's own fine depths and deformed points."""
    from oracle import stnerf_oracle as O
    from stnerf_amd import synthetic as syn
    L = meta["L"]
    fl = meta.get("flags", {})
    sd = {k: v.to(dtype).requires_grad_(True) for k, v in syn.state_dict_for_flags(L, meta["space_time"], meta["deform_time"],
                                                                                  meta["weight_seed"], fl).items()}
    bk, per = syn.scene_boxes(L)
    m = O.OracleModel(layer_num=L, n_coarse=meta["n1"], n_fine=meta["n2"], params=sd, use_deform_time=meta["deform_time"],
                      use_space_time=meta["space_time"], bkgd_use_deform_time=fl.get("BKGD_USE_DEFORM_TIME", False),
                      bkgd_use_space_time=fl.get("BKGD_USE_SPACE_TIME", False), bkgd_bbox=bk.to(dtype), bboxes=per.to(dtype))
    draws, rp = replay_of(z, meta, dtype)
    it = iter(draws)
    forced = {k: rp[k] for k in ("z", "xyz_c", "xyz_f")} if teacher else None
    out = O.render_chunk(m, torch.from_numpy(z["rays"]).to(dtype), only_coarse=meta["only_coarse"], rand=lambda shape: next(it),
                         sample_dtype=sample_dtype, forced=forced)
    loss, parts = trainer_loss(out, torch.from_numpy(z["rgbs"]).to(dtype), torch.from_numpy(z["labels"]), meta["only_coarse"],
                               meta["remove_outliers"])
    loss.backward()
    return sd, out, loss, parts
