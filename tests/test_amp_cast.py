"""amp_cast.CastMany: the dense head's reduced-precision parameter copies of a training step (one multi-tensor launch each way)."""
import torch

from mv3d_tf_amd.amp_cast import CastMany, cast_params


def test_cast_many_values_and_gradients_equal_per_tensor_casts():
    torch.manual_seed(0)
    params = {"a": [torch.randn(5, 7, requires_grad=True), torch.randn(5, requires_grad=True)],
              "b": [torch.randn(3, 5, requires_grad=True), torch.randn(3, requires_grad=True)]}
    half = cast_params(params, ["a", "b"], torch.bfloat16)
    for n in params:
        for got, p in zip(half[n], params[n]):
            assert got.dtype == torch.bfloat16 and torch.equal(got, p.detach().to(torch.bfloat16))
    x = torch.randn(4, 7).to(torch.bfloat16)
    y = torch.nn.functional.linear(torch.nn.functional.linear(x, *half["a"]), *half["b"])
    y.float().square().sum().backward()
    # the same step with one cast per tensor (what autocast does)
    ref = {n: [p.detach().clone().requires_grad_(True) for p in ps] for n, ps in params.items()}
    y2 = torch.nn.functional.linear(torch.nn.functional.linear(x, *[p.to(torch.bfloat16) for p in ref["a"]]), *[p.to(torch.bfloat16) for p in ref["b"]])
    y2.float().square().sum().backward()
    for n in params:
        for p, q in zip(params[n], ref[n]):
            assert p.grad.dtype == torch.float32 and torch.equal(p.grad, q.grad)


def test_cast_many_skips_tensors_without_a_gradient():
    w, b = torch.randn(3, 3, requires_grad=True), torch.randn(3, requires_grad=True)
    hw, hb = CastMany.apply(torch.bfloat16, w, b)
    hw.float().sum().backward()
    assert b.grad is None and torch.equal(w.grad, torch.ones(3, 3))
