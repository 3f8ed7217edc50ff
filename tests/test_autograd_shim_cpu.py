"""Host logic of the autograd shim (ecg_denoise_amd.model._AutogradMixin / _RalFunction) on a CPU stand-in: a toy "library"
with flat parameter / gradient buffers whose backward OVERWRITES the gradient buffer, exactly like ral_backward.  The GPU
counterpart (the reference loop on the real engine) is tests/test_gpu_reference_loop.py."""
import pytest
import torch
import torch.nn.functional as F

from ecg_denoise_amd import RalError
from ecg_denoise_amd.model import _AutogradMixin


class Toy(_AutogradMixin):
    """y = x * a + b (a, b: vectors of the window length), parameters in ONE flat buffer"""

    def __init__(self, n):
        self.n, self.training = n, True
        self.params = torch.cat([torch.full((n,), 2.0), torch.zeros(n)])
        self.grads = torch.zeros(2 * n)
        self.calls = 0
        self.enable_autograd(True)

    def _flat_pairs(self):
        n = self.n
        return [("a", self.params[:n], self.grads[:n]), ("b", self.params[n:], self.grads[n:])]

    def parameters(self):
        return [leaf for _, leaf, _ in self._leaf_parameters()]

    def _forward_raw(self, x):
        self._x = x.detach()
        return self._x * self.params[:self.n] + self.params[self.n:]

    def _backward_raw(self, dy, want_dx):
        self.calls += 1
        self.grads.zero_()                                     # (ral_backward zeroes, then writes)
        self.grads[:self.n] += (dy * self._x).sum(0)
        self.grads[self.n:] += dy.sum(0)
        return dy * self.params[:self.n] if want_dx else None

    def __call__(self, x):
        return self._forward_autograd(x)


def test_reference_loop_runs_unchanged_on_the_shim():
    torch.manual_seed(0)
    n, B = 8, 4
    m = Toy(n)
    a = torch.full((n,), 2.0, requires_grad=True); b = torch.zeros(n, requires_grad=True)
    opt_m, opt_r = torch.optim.Adam(m.parameters(), lr=0.001), torch.optim.Adam([a, b], lr=0.001)
    for step in range(4):
        data, target = torch.randn(B, n), torch.randn(B, n)
        opt_m.zero_grad(); opt_r.zero_grad()                   # (set_to_none=True: .grad detached from the flat buffer)
        pre = m(data)
        assert pre.grad_fn is not None
        loss = F.mse_loss(pre, target)
        loss.backward()
        lr_ = F.mse_loss(data * a + b, target)
        lr_.backward()
        leaves = m.parameters()
        assert leaves[0].grad.data_ptr() == m.grads.data_ptr()            # .grad aliases the flat gradient buffer
        torch.testing.assert_close(leaves[0].grad, a.grad); torch.testing.assert_close(leaves[1].grad, b.grad)
        opt_m.step(); opt_r.step()
        torch.testing.assert_close(m.params[:n], a.detach()); torch.testing.assert_close(m.params[n:], b.detach())
        assert abs(loss.item() - lr_.item()) < 1e-6
    assert m.calls == 4


def test_gradients_accumulate_when_the_caller_does_not_zero_them():
    n = 4
    m = Toy(n)
    x = torch.ones(2, n)
    m(x).sum().backward()
    g1 = m.parameters()[0].grad.clone()
    m(x).sum().backward()                                       # no zero_grad in between: torch semantics = accumulate
    torch.testing.assert_close(m.parameters()[0].grad, 2 * g1)
    for p in m.parameters():
        p.grad.zero_()                                          # zero_grad(set_to_none=False)
    m(x).sum().backward()
    torch.testing.assert_close(m.parameters()[0].grad, g1)
    # a gradient tensor of the caller's own is accumulated into, not replaced
    own = torch.ones(n)
    m.parameters()[0].grad = own
    m(x).sum().backward()
    assert m.parameters()[0].grad is own
    torch.testing.assert_close(own, 1 + g1)


def test_input_gradient_and_stale_forward():
    n = 4
    m = Toy(n)
    x = torch.ones(2, n, requires_grad=True)
    m(x).sum().backward()
    torch.testing.assert_close(x.grad, torch.full((2, n), 2.0))
    y_old = m(torch.ones(2, n))
    m(torch.ones(2, n))                                         # a newer forward replaced the library's saved activations
    with pytest.raises(RalError, match="most recent"):
        y_old.sum().backward()
