"""The grasp heads in TRAINING mode as one autograd node each (csrc/heads_train.hip, heads_train.py) against the
layer-by-layer torch path (conv + BatchNorm1d on batch statistics + ReLU: pointnet2.py:174-188, :240-253) and against the same
modules in float64: outputs, running statistics, counters, every parameter gradient and the input gradient.
Tolerance: the native path may be at most 3 x as far from float64 as torch's own fp32 path, plus 1e-5 of the tensor's scale (at least 1)
(both are fp32 GEMMs with different summation orders followed by a normalisation)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _heads(kind, seed):
    from regnet_for_3d_grasping_amd import synthetic
    from regnet_for_3d_grasping_amd.pointnet2 import PointNet2Refine, PointNet2TwoStage
    m = PointNet2TwoStage(256, 6, 4, 40, 4) if kind == "two" else PointNet2Refine(64, 6, 2, 10)
    m.load_state_dict(synthetic.seeded_state_dict(m, seed))
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.8 + 0.6)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.2)
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    return m.to(DEV).train()


def _run(kind, module, x, extra, wo):
    """forward + backward of a head; returns (outputs, input grad, {name: param grad}, {name: buffer})."""
    x = x.clone().requires_grad_(True)
    if kind == "two":
        c, r, _ = module(x, None, pooled=True)
    else:
        c, r = module(x, extra, pooled=True)
    loss = (c * wo[0]).sum() + (r * wo[1].view_as(r)).sum()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in module.named_parameters() if p.grad is not None}
    bufs = {n: b.detach().clone() for n, b in module.named_buffers()}
    return (c.detach(), r.detach()), x.grad.detach().clone(), grads, bufs


@pytest.mark.parametrize("kind,rows", [("two", 512), ("two", 450), ("two", 37), ("ref", 444), ("ref", 19), ("ref", 2)])
def test_heads_train_node_matches_torch_layers(kind, rows, monkeypatch):
    from regnet_for_3d_grasping_amd import heads_train
    base = _heads(kind, 11 + rows)
    g = torch.Generator().manual_seed(rows)
    if kind == "two":
        x = torch.randn(rows, 256, 1, generator=g).to(DEV)
        extra = None
        wo = (torch.randn(rows, 4, generator=g).to(DEV), torch.randn(rows, 40, generator=g).to(DEV))
    else:
        x = torch.randn(rows, 256, 1, generator=g).to(DEV)
        extra = torch.randn(rows, 128, generator=g).to(DEV)
        wo = (torch.randn(rows, 2, generator=g).to(DEV), torch.randn(rows, 10, generator=g).to(DEV))
    res = {}
    for name, enabled, dtype in (("native", True, torch.float32), ("torch", False, torch.float32), ("f64", False, torch.float64)):
        m = copy.deepcopy(base).to(dtype)
        monkeypatch.setattr(heads_train, "ENABLED", enabled)
        before = dict(heads_train.CALLS)
        res[name] = _run(kind, m, x.to(dtype), None if extra is None else extra.to(dtype), tuple(w.to(dtype) for w in wo))
        used = heads_train.CALLS["forward"] - before["forward"], heads_train.CALLS["backward"] - before["backward"]
        assert used == ((1, 1) if enabled else (0, 0)), (name, used)

    def check(what, a, b, ref):
        ref = ref.double()
        scale = max(1.0, float(ref.abs().max()))
        e_native, e_torch = float((a.double() - ref).abs().max()), float((b.double() - ref).abs().max())
        assert e_native <= 3.0 * e_torch + 1e-5 * scale, (what, e_native, e_torch, scale)

    for k in range(2):
        check("out%d" % k, res["native"][0][k], res["torch"][0][k], res["f64"][0][k])
    check("dx", res["native"][1], res["torch"][1], res["f64"][1])
    assert set(res["native"][2]) == set(res["torch"][2])
    for n in res["f64"][2]:
        if n.startswith("conv") and n.endswith(".bias"):
            # a convolution bias in front of a BatchNorm has a zero gradient; both fp32 paths return rounding noise
            assert float(res["native"][2][n].abs().max()) <= 1e-4 * max(1.0, float(res["native"][1].abs().max()))
            continue
        check(n, res["native"][2][n], res["torch"][2][n], res["f64"][2][n])
    for n in res["f64"][3]:
        if n.endswith("num_batches_tracked"):
            assert int(res["native"][3][n]) == int(res["torch"][3][n]) == int(res["f64"][3][n])
        else:
            check(n, res["native"][3][n], res["torch"][3][n], res["f64"][3][n])


def test_head_layer_train_abi_rejects_bad_shapes():
    from regnet_for_3d_grasping_amd import _lib
    L = _lib.lib
    assert L.regnet_head_layer_train_supported(512, 256, 1024) == 1
    assert L.regnet_head_layer_train_supported(1, 256, 4) == 0          # BatchNorm needs two rows
    assert L.regnet_head_layer_train_supported(2000, 256, 4) == 0
    assert L.regnet_head_layer_train_supported(512, 100, 4) == 0
    x = torch.zeros(4, 64, device=DEV)
    assert L.regnet_head_layer_train_fwd_f32(x.data_ptr(), 64, None, None, None, None, None, None, None, 0.1, 1e-5, 4, 64, 8, 1,
                                             None, None, None, None) != 0
