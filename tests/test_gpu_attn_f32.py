"""fp32 attention on the matrix cores (psg_attn_f32.hip; `-m gpu`): the kernels behind the reference-precision modes
(head dtype 'fp32' / 'fp32s') against fp64 torch restatements of HF-IB:464-515 / HF-LL:191-214 and against the scalar
checker kernels they replace, on the cases the goldens do not reach: empty mask unions under both policies, ragged
pair lists, patch counts that are not multiples of 64, prompts of every length, NaN-poisoned unwritten cache rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _bits(om, L):
    N = om.shape[0]
    words = (L + 63) // 64
    b = np.zeros((N, words * 64), dtype=np.uint8)
    b[:, :L] = om.numpy()
    return torch.from_numpy(np.packbits(b, axis=-1, bitorder="little").view(np.int64).reshape(N, words)).to(DEV)


def _xattn_fp64(q, k, v, pm, heads, nq, uniform):
    L = k.shape[0]
    qh = q.double().view(-1, nq, heads, 64).permute(0, 2, 1, 3)
    kh = k.double().view(L, heads, 64).permute(1, 0, 2)
    vh = v.double().view(L, heads, 64).permute(1, 0, 2)
    s = torch.einsum("phqd,hld->phql", qh, kh) * 0.125
    on = pm[:, None, None, :]
    empty = ~pm.any(-1)
    if uniform:
        s = torch.where(on, s, torch.full_like(s, float("-inf")))
        s[empty] = 0.0                                              # finfo.min absorbs every score: uniform over L keys
    else:
        s = torch.where(on, s, s - 10000.0)                        # legacy additive mask: an empty union = unmasked
    o = torch.einsum("phql,hld->phqd", torch.softmax(s, -1), vh)
    return o.permute(0, 2, 1, 3).reshape(-1, heads * 64)


@pytest.mark.parametrize("L,N,nq,policy,rect", [
    (256, 50, 33, "uniform", True), (256, 50, 1, "uniform", True), (256, 12, 33, "unmasked", False),
    (336, 9, 33, "uniform", True), (336, 9, 1, "unmasked", False), (64, 5, 33, "uniform", False),
    (192, 7, 33, "uniform", False), (40, 3, 33, "unmasked", False), (100, 6, 1, "uniform", False),
    (256, 10, 17, "uniform", True), (256, 10, 49, "uniform", True)])
def test_cross_attn_f32_vs_fp64_and_scalar_kernel(L, N, nq, policy, rect):
    from openpsg_amd import _lib, ops
    g = torch.Generator(device="cpu").manual_seed(L * 1000 + N * 10 + nq)
    heads = 12
    if rect:                                                       # object rectangles on the patch grid, as real masks
        gw = 16 if L == 256 else 21
        gh = L // gw
        om = torch.zeros(N, gh, gw, dtype=torch.bool)
        for i in range(N):
            y0, x0 = int(torch.randint(0, gh - 1, (1,), generator=g)), int(torch.randint(0, gw - 1, (1,), generator=g))
            h_, w_ = int(torch.randint(1, 7, (1,), generator=g)), int(torch.randint(1, 7, (1,), generator=g))
            om[i, y0:y0 + h_, x0:x0 + w_] = True
        om = om.reshape(N, L)
    else:
        om = torch.rand(N, L, generator=g) < 0.12
    om[0] = False                                                  # object 0 vanished -> pair (0, 0) is empty
    if N > 4:
        om[N - 1] = False
    perm = torch.randperm(N * N, generator=g)
    pair_index = torch.cat([torch.tensor([0]), perm[: max(1, (N * N * 3) // 4)]]).to(torch.int32)   # shuffled subset, pair 0 first
    P = pair_index.numel()
    q = (torch.randn(P * nq, 768, generator=g) * 1.5).to(DEV)
    k = (torch.randn(L, 768, generator=g) * 1.5).to(DEV)
    v = torch.randn(L, 768, generator=g).to(DEV)
    bits = _bits(om, L)
    pm = (om[:, None, :] | om[None, :, :]).reshape(N * N, L)[pair_index.long()].to(DEV)
    pol = _lib.PSG_EMPTY_UNIFORM if policy == "uniform" else _lib.PSG_EMPTY_UNMASKED
    ref = _xattn_fp64(q, k, v, pm, heads, nq, policy == "uniform")
    out = torch.full_like(q, float("nan"))
    ops.qformer_cross_attn(q, k, v, bits, pair_index.to(DEV), N, nq, heads, out=out, empty_policy=pol)   # default = f32 MFMA
    chk = ops.qformer_cross_attn(q, k, v, bits, pair_index.to(DEV), N, nq, heads, empty_policy=pol,
                                 variant=_lib.PSG_XATTN_SIMPLE)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    e_m = (out.double() - ref).abs().max().item()
    e_s = (chk.double() - ref).abs().max().item()
    print(f"L={L} N={N} nq={nq} {policy}: P={P}, f32 MFMA err {e_m:.2e}, scalar kernel err {e_s:.2e}")
    # the scalar checker adds the legacy -10000 in fp32 (as HF's legacy mask does): on an EMPTY union every score is
    # rounded to the ulp of 10000 (1e-3) before the softmax; the matrix-core kernel scores an empty union unmasked, exactly
    assert e_m < 1e-5 and e_s < (2e-5 if policy == "uniform" else 2e-3)
    if policy == "uniform":                                        # empty pair (0, 0): the mean of V
        assert (out[:nq].double() - v.double().mean(0)[None]).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,T,q_only", [(7, 14, 0), (5, 17, 1), (3, 31, 0), (4, 9, 1), (9, 14, 2), (2, 0, 0), (6, 30, 1)])
def test_self_attn_f32_vs_fp64(B, T, q_only):
    from openpsg_amd import _lib, ops
    g = torch.Generator().manual_seed(B * 100 + T)
    nq, heads, H = 33, 12, 768
    R = B * (nq + T)
    qkv = (torch.randn(R, 3 * H, generator=g) * 1.2).to(DEV)
    tmask = (torch.rand(B, max(T, 1), generator=g) < 0.8).to(torch.uint8)[:, :T]
    if T:
        tmask[:, 0] = 1
        tmask[0] = 0                                               # a pair whose prompt is fully masked: query rows only
    tmask = tmask.contiguous().to(DEV)
    out = torch.full((R, H), float("nan"), device=DEV)
    if q_only == 2:
        # the cls-only mode is reached through the ABI directly (compact output, one row per pair)
        lib, ctx = _lib.load(), _lib.ctx(0)
        outc = torch.full((B, H), float("nan"), device=DEV)
        _lib.check(lib.psg_qformer_self_attn(ctx, qkv.data_ptr(), tmask.data_ptr(), B, T, nq, heads, 2, outc.data_ptr(),
                                             _lib.PSG_F32, torch.cuda.current_stream().cuda_stream), "self_attn")
    else:
        ops.qformer_self_attn(qkv, tmask, B, T, nq, heads, bool(q_only), out)
    f = qkv.double()
    worst = 0.0
    for p in range(B):
        rows = list(range(p * nq, (p + 1) * nq)) + list(range(B * nq + p * T, B * nq + (p + 1) * T))
        x = f[rows]
        q_, k_, v_ = [x[:, i * H:(i + 1) * H].view(-1, heads, 64).permute(1, 0, 2) for i in range(3)]
        valid = torch.cat([torch.ones(nq, device=DEV, dtype=torch.bool), tmask[p].bool()])
        s = (q_ @ k_.transpose(1, 2) * 0.125).masked_fill(~valid[None, None, :], float("-inf"))
        o = (torch.softmax(s, -1) @ v_).permute(1, 0, 2).reshape(-1, H)
        if q_only == 2:
            worst = max(worst, (outc[p].double() - o[0]).abs().max().item())
            continue
        n = nq if q_only else nq + T
        got = out[rows][:n].double()
        assert torch.isfinite(got).all()
        worst = max(worst, (got - o[:n]).abs().max().item())
    print(f"self_attn f32 B={B} T={T} q_only={q_only}: max err vs fp64 {worst:.2e}")
    assert worst < 1e-5


@pytest.mark.parametrize("B,T", [(7, 14), (33, 9), (4, 31)])
def test_self_attn_f32_shared_query_rows_equals_full_kernel(B, T):
    """Layer 0 in fp32: the shared-block entry against the full kernel on a qkv matrix that repeats the block per pair
    (same arithmetic per unit: bit-identical)."""
    from openpsg_amd import ops
    g = torch.Generator().manual_seed(B * 10 + T)
    nq, heads, H = 33, 12, 768
    qkv_q = torch.randn(nq, 3 * H, generator=g).to(DEV)
    qkv_t = torch.randn(B * T, 3 * H, generator=g).to(DEV)
    mask = (torch.rand(B, T, generator=g) < 0.8).to(torch.uint8)
    mask[:, 0] = 1
    mask = mask.to(DEV)
    full = torch.cat([qkv_q.repeat(B, 1), qkv_t]).contiguous()
    want = torch.empty(B * (nq + T), H, device=DEV)
    ops.qformer_self_attn(full, mask, B, T, nq, heads, False, want)
    got = torch.full_like(want, 5.0)
    ops.qformer_self_attn_shared(qkv_q, qkv_t, mask, B, T, nq, heads, got)
    assert torch.equal(got, want)


@pytest.mark.parametrize("K,S,heads,extra", [(20, 48, 32, 16), (3, 64, 4, 0), (5, 33, 2, 3), (1, 7, 1, 1), (4, 17, 3, 40)])
def test_prefill_attn_f32_vs_fp64_and_scalar_kernel(K, S, heads, extra):
    """psg_prefill_attn with fp32 activations against an fp64 causal attention and psg_llm_attn (scalar) on a pair-major
    prompt batch with ragged lengths; cache rows the rotary kernel never wrote are NaN."""
    from openpsg_amd import ops
    g = torch.Generator(device="cpu").manual_seed(K * 100 + S)
    D, ctx = heads * 128, S + extra
    lens = torch.randint(max(1, S - 20), S + 1, (K,), generator=g)
    lens[0] = S
    if K > 2:
        lens[1] = 1
        lens[2] = 16
    q = torch.randn(K * S, D, generator=g).to(DEV)
    kc = torch.zeros(K, heads, ctx, 128, device=DEV)
    vc = torch.zeros_like(kc)
    kc[:, :, :S] = torch.randn(K, heads, S, 128, generator=g).to(DEV)
    vc[:, :, :S] = torch.randn(K, heads, S, 128, generator=g).to(DEV)
    t = torch.arange(S)[None, :].expand(K, -1)
    pos = torch.where(t < lens[:, None], t, torch.full_like(t, -1)).reshape(-1).to(torch.int32).to(DEV)
    pair = torch.arange(K, dtype=torch.int32)[:, None].expand(-1, S).reshape(-1).contiguous().to(DEV)
    kc_clean, vc_clean = kc.clone(), vc.clone()
    for kk in range(K):
        kc[kk, :, int(lens[kk]):] = float("nan")
        vc[kk, :, int(lens[kk]):] = float("nan")
    out_m = torch.full((K * S, D), 7.0, device=DEV)
    out_s = torch.empty_like(out_m)
    ops.prefill_attn(q, kc, vc, pos, K, S, heads, 128, ctx, out_m)
    ops.llm_attn(q, kc, vc, pair, pos, heads, 128, ctx, out_s)
    qh = q.double().view(K, S, heads, 128).permute(0, 2, 1, 3)
    sc = torch.einsum("khqd,khjd->khqj", qh, kc_clean[:, :, :S].double()) / 128 ** 0.5
    causal = torch.ones(S, S, dtype=torch.bool, device=DEV).tril()
    keyok = (t < lens[:, None]).to(DEV)
    sc = sc.masked_fill(~(causal[None, None] & keyok[:, None, None, :]), float("-inf"))
    ref = torch.einsum("khqj,khjd->khqd", torch.softmax(sc, -1), vc_clean[:, :, :S].double()).permute(0, 2, 1, 3).reshape(K * S, D)
    ok = pos >= 0
    e_m = (out_m.double() - ref)[ok].abs().max().item()
    e_s = (out_s.double() - ref)[ok].abs().max().item()
    print(f"K={K} S={S}: f32 MFMA err {e_m:.2e}, scalar err {e_s:.2e}")
    assert torch.isfinite(out_m).all() and e_m < 1e-5 and e_s < 1e-5
    assert (out_m[~ok] == 0).all()                                  # padding rows: zeros, like the scalar kernel
