"""sharedLayers operator API (reference signatures) on the GPU: values and registered gradients."""
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def test_correlation_op_and_gradient(hip):
    from Nets import sharedLayers as SL
    x = torch.randn(2, 12, 40, 128, device="cuda", requires_grad=True)
    y = torch.randn(2, 12, 40, 128, device="cuda", requires_grad=True)
    out = SL.correlation(x, y, 2, stride=1)
    xc = x.detach().cpu().requires_grad_(True); yc = y.detach().cpu().requires_grad_(True)
    ref = T.correlation(xc, yc, 2, 1)
    assert (out.cpu() - ref).abs().max().item() < 1e-5
    g = torch.randn_like(out)
    out.backward(g)
    gx, gy = torch.autograd.grad(ref, [xc, yc], g.cpu())
    assert (x.grad.cpu() - gx).abs().max().item() < 1e-5 and (y.grad.cpu() - gy).abs().max().item() < 1e-5
    # mode='TF' (the reference's default) and mode='CUDA' name the same formula: both run the HIP op here, bit-identical to the default call
    for mode in ('TF', 'CUDA'):
        assert torch.equal(SL.correlation(x, y, 2, mode=mode), out)
    with pytest.raises(Exception):
        SL.correlation(x, y, 2, mode='numpy')


def test_conv_ops_and_gradients(hip):
    from Nets import sharedLayers as SL
    st = SL.VariableStore(device="cuda", seed=3)
    x = torch.randn(1, 24, 40, 16, device="cuda", requires_grad=True)
    y1 = SL.conv2d(x, [3, 3, 16, 32], strides=2, activation=SL.Leaky(0.2), name='c1', bName='biases', store=st)
    y2 = SL.dilated_conv2d(y1, [3, 3, 32, 32], rate=2, activation=SL.Leaky(0.2), name='c2', store=st)
    y3 = SL.conv2d_transpose(y2, [4, 4, 8, 32], strides=2, name='d1', store=st)
    loss = (y3 ** 2).mean()
    loss.backward()
    # oracle
    w = {k: v.detach().cpu().requires_grad_(True) for k, v in st.vars.items()}
    xc = x.detach().cpu().requires_grad_(True)
    r1 = T.conv2d(xc, w['c1/weights'], w['c1/biases'], stride=2, alpha=0.2)
    r2 = T.conv2d(r1, w['c2/weights'], w['c2/biases'], dilation=2, alpha=0.2)
    r3 = T.conv2d_transpose(r2, w['d1/weights'], w['d1/bias'], stride=2, alpha=0.1)
    lr = (r3 ** 2).mean()
    lr.backward()
    assert abs(loss.item() - lr.item()) < 1e-5 * max(1, abs(lr.item()))
    assert (x.grad.cpu() - xc.grad).abs().max().item() < 1e-4 * max(1.0, xc.grad.abs().max().item())
    for k in w:
        assert (st.vars[k].grad.cpu() - w[k].grad).abs().max().item() < 2e-4 * max(1e-3, w[k].grad.abs().max().item()), k
