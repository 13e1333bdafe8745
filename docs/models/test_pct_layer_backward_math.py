"""The hand-written backward pass of ops.OffsetAttentionLayer (PCT's offset-attention layer, model/pct_cls.py:111-139), formula by
formula in float64 on the CPU against autograd of the forward composition: the gradients of v and of the attention are carried
NEGATED (the normalisation's backward is linear), the signs come back through addmm(alpha = -1), the residual's gradient is the first
addmm's addend.  The GPU test checks the kernels; this one checks the algebra."""
import torch


def oa_norm(E):
    S = torch.softmax(E, dim=-1)
    return S / (1e-9 + S.sum(dim=1, keepdim=True))


def oa_norm_backward(dA, E):
    """d E of A = oa_norm(E) by autograd (linear in dA)."""
    E = E.detach().requires_grad_()
    A = oa_norm(E)
    g, = torch.autograd.grad(A, E, dA)
    return g


def test_hand_written_backward_equals_autograd():
    torch.manual_seed(4)
    B, N, C, Cq = 3, 64, 32, 8
    x = torch.randn(B, N, C, dtype=torch.float64, requires_grad=True)
    Wq, Wv, Wt = (torch.randn(Cq, C, dtype=torch.float64) * 0.3, torch.randn(C, C, dtype=torch.float64) * 0.2,
                  torch.randn(C, C, dtype=torch.float64) * 0.2)
    bv, bt = torch.randn(C, dtype=torch.float64) * 0.1, torch.randn(C, dtype=torch.float64) * 0.1
    g = torch.randn(B, N, C, dtype=torch.float64)
    # forward, as SA_Layer.forward_pm composes it
    q = x @ Wq.t()
    E = q @ q.transpose(1, 2)
    A = oa_norm(E)
    v = x @ Wv.t() + bv
    x_r = A.transpose(1, 2) @ v
    y = torch.relu((x - x_r) @ Wt.t() + bt)
    out = x + y
    want, = torch.autograd.grad(out, x, g)
    # backward, as OffsetAttentionLayer.backward writes it
    with torch.no_grad():
        g2 = g.reshape(B * N, C)
        dt = (g2 * (y.reshape(B * N, C) > 0)) @ Wt
        dt3 = dt.view(B, N, C)
        dv_neg = A @ dt3
        dA_neg = v @ dt3.transpose(1, 2)
    dE_neg = oa_norm_backward(dA_neg, E)
    with torch.no_grad():
        dq_neg = dE_neg @ q + dE_neg.transpose(1, 2) @ q
        dx = torch.addmm(g2, dv_neg.reshape(B * N, C), Wv, alpha=-1.0)
        dx = torch.addmm(dx, dq_neg.reshape(B * N, Cq), Wq, alpha=-1.0)
        dx += dt
    assert torch.allclose(dx.view(B, N, C), want, rtol=1e-10, atol=1e-12)
