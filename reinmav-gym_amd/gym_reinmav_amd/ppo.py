"""PPO2-style rollout collection (and a minimal learner) on top of the batched HIP envs.

This is the *caller* side of the hot path: what ``python -m gym_reinmav.run --alg=ppo2 --env=quadrotor3d-v0
--network=mlp`` does through baselines (``gym_reinmav/run.py:63-68``: ``learn(env=...)`` whose ``Runner``
alternates ``model.step(obs)`` and ``env.step(actions)``, then GAE(lambda) and clipped-surrogate
minibatch epochs).  baselines / TensorFlow 1 are third party and absent; the hyper-parameters below are
baselines' documented ppo2 defaults (nsteps 2048 for ONE env - here the batch is 10^4-10^6 envs, so a
rollout is ``nsteps`` x N transitions with a small ``nsteps``).

MI355X-first choices:
* everything stays on the GPU: observations never leave HBM, ``env.step`` receives the action tensor's
  device pointer and writes obs / reward / done straight into the rollout buffers (zero copies);
* **feature-major layout end to end**: the env's native layout is SoA ``[dim, N]``; the policy computes
  ``W @ X`` on ``X = obs[nS, N]`` and emits actions ``[nA, N]``, so neither side ever transposes and
  every env access stays coalesced (no AoS emit needed);
* the whole T-step rollout (policy + env launches) can be captured in ONE hipGraph
  (``RolloutCollector(graph=True)``): ``rmav_step`` with device pointers allocates nothing and never
  synchronises, so it is capturable; replay removes ~10 eager launches of host overhead per env-step;
* data parallel over GPUs = env sharding; gradients are averaged with one flat all-reduce per minibatch
  (~10 k parameters: latency-bound on xGMI, so a single bucket).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist

from .core import BatchedQuadrotor


class _LinearFM(torch.autograd.Function):
    """``W @ x + b[:, None]`` on feature-major activations ``x [in, B]`` with a weight gradient that suits the learner's
    shapes.  ``dW = dy @ x^T`` is a 64 x 64 result reduced over B = 524 288 samples: the BLAS library's pick for that
    long-K product ran 355 us per call - 45 % of a PPO iteration at 65 536 envs x 32 steps.  Cut into chunks of
    ``CHUNK`` columns it is a batched 64 x 64 x CHUNK product over hundreds of workgroups plus a small sum."""

    CHUNK = 8192
    _ones = None   # one vector of ones, grown to the largest batch seen (sliced for smaller ones)

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias[:, None], weight, x)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = weight.t() @ gy if ctx.needs_input_grad[0] else None
        B, C = x.shape[1], _LinearFM.CHUNK
        if B % C == 0 and B > C:
            nb = B // C
            # autograd may hand over a non-contiguous gy (expanded / transposed grads), callers a permuted x
            gy_c, x_c = gy.contiguous(), x.contiguous()
            gw = torch.bmm(gy_c.view(gy.shape[0], nb, C).transpose(0, 1),          # [nb, out, C]
                           x_c.view(x.shape[0], nb, C).permute(1, 2, 0)).sum(0)     # [nb, C, in] -> [out, in]
        else:
            gw = gy @ x.t()
        # bias gradient as a matrix-vector product with ones (the generic row reduction ran at 2.7 TB/s)
        ones = _LinearFM._ones
        if ones is None or ones.numel() < B or ones.device != gy.device or ones.dtype != gy.dtype:
            ones = _LinearFM._ones = torch.ones(B, dtype=gy.dtype, device=gy.device)
        return gx, gw, torch.mv(gy, ones[:B])


class MlpPolicy(torch.nn.Module):
    """baselines ``mlp`` (2 x 64 tanh) Gaussian policy, in feature-major form.

    ``value_network``: ``'copy'`` (default) = a separate value net of the same shape - baselines' MuJoCo default and its classic
    ``MlpPolicy``; ``'shared'`` = a scalar value head on the policy's latent - what baselines' ``build_policy`` does when
    ``value_network`` is None, i.e. for an env type without a ppo2 defaults entry such as the native envs of gym_reinmav
    (``run.py:63-68``; third-party behaviour restated from memory).  ``self.vf`` is then the one-layer head."""

    def __init__(self, n_obs: int, n_act: int, hidden: int = 64, init_logstd: float = 0.0, value_network: str = "copy"):
        super().__init__()
        assert value_network in ("copy", "shared")
        self.shared = value_network == "shared"
        mk = lambda i, o: torch.nn.Linear(i, o)  # noqa: E731
        self.pi = torch.nn.ModuleList([mk(n_obs, hidden), mk(hidden, hidden), mk(hidden, n_act)])
        self.vf = torch.nn.ModuleList([mk(hidden, 1)] if self.shared else [mk(n_obs, hidden), mk(hidden, hidden), mk(hidden, 1)])
        self.logstd = torch.nn.Parameter(torch.full((n_act,), float(init_logstd)))
        for net, last_gain in ((self.pi, 0.01), (self.vf, 1.0)):
            for i, lin in enumerate(net):
                torch.nn.init.orthogonal_(lin.weight, gain=last_gain if lin is net[-1] else math.sqrt(2.0))
                torch.nn.init.zeros_(lin.bias)

    @staticmethod
    def _mlp(net, x):  # x [in, N] -> [out, N]
        for i, lin in enumerate(net):
            x = (_LinearFM.apply(x, lin.weight, lin.bias) if torch.is_grad_enabled() and x.is_cuda
                 else torch.addmm(lin.bias[:, None], lin.weight, x))
            if i < 2:
                x = torch.tanh(x)
        return x

    def _lin(self, lin, x):
        return (_LinearFM.apply(x, lin.weight, lin.bias) if torch.is_grad_enabled() and x.is_cuda
                else torch.addmm(lin.bias[:, None], lin.weight, x))

    def forward(self, obs_fm: torch.Tensor):
        """obs_fm [nS, N] -> (mean [nA, N], value [N])."""
        if self.shared:
            h = torch.tanh(self._lin(self.pi[1], torch.tanh(self._lin(self.pi[0], obs_fm))))
            return self._lin(self.pi[2], h), self._lin(self.vf[0], h)[0]
        return self._mlp(self.pi, obs_fm), self._mlp(self.vf, obs_fm)[0]

    def log_prob(self, mean, act):
        z = (act - mean) * torch.exp(-self.logstd)[:, None]
        return -0.5 * (z * z).sum(0) - self.logstd.sum() - 0.5 * mean.shape[0] * math.log(2 * math.pi)

    def entropy(self):
        return (self.logstd + 0.5 * math.log(2 * math.pi * math.e)).sum()


class RolloutCollector:
    """Collects ``nsteps`` transitions of every env into device-resident, time-major, feature-major buffers."""

    def __init__(self, env: BatchedQuadrotor, policy: MlpPolicy, nsteps: int, graph: bool = False):
        assert env.auto_reset, "rollouts need VecEnv semantics (auto-reset)"
        self.env, self.policy, self.T = env, policy, int(nsteps)
        dev = torch.device("cuda", env.device)
        N, nS, nA, T = env.num_envs, env.nS, env.nA, self.T
        f32 = dict(dtype=torch.float32, device=dev)
        self.obs = torch.empty((T + 1, nS, N), **f32)   # obs[t] is the observation action t was computed from
        self.act = torch.empty((T, nA, N), **f32)
        self.logp = torch.empty((T, N), **f32)
        self.val = torch.empty((T + 1, N), **f32)
        self.rew = torch.empty((T, N), **f32)
        self.done = torch.empty((T, N), dtype=torch.uint8, device=dev)
        self.obs[0].copy_(env.get_state(layout="soa", device_out=True))
        self._graph = None
        if graph:
            self._capture()

    # one env-step of the loop baselines' Runner.run() executes (model.step -> env.step)
    def _step(self, t: int):
        mean, v = self.policy(self.obs[t])
        noise = torch.randn_like(mean)
        torch.addcmul(mean, noise, torch.exp(self.policy.logstd)[:, None], out=self.act[t])
        self.logp[t] = -0.5 * (noise * noise).sum(0) - self.policy.logstd.sum() - 0.5 * mean.shape[0] * math.log(2 * math.pi)
        self.val[t] = v
        self.env.step(self.act[t], layout="soa", out=(self.obs[t + 1], self.rew[t], self.done[t]))

    def _body(self):
        with torch.no_grad():
            for t in range(self.T):
                self._step(t)
            self.val[self.T] = self.policy(self.obs[self.T])[1]

    def _capture(self):
        env = self.env
        side = torch.cuda.Stream(device=env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        state = env.get_state(layout="soa", device_out=True)
        t0 = env.step_count
        with torch.cuda.stream(side):
            # warm-up (allocator, lazy inits of the torch ops) on a scratch env of the same shape, so that the
            # caller's envs - state, steps_beyond_done, reset counters, episode accumulators and totals, clocks -
            # are not advanced by it; the capture below records launches on the real env without executing them
            tmp = BatchedQuadrotor(env.kind, env.num_envs, device=env.device, seed=0, auto_reset=True,
                                   track_episodes=env.track_episodes, params=env.params)
            self.env = tmp
            try:
                self._body()
            finally:
                self.env = env
            side.synchronize()
            tmp.close()
            env.use_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                self._body()
        self._graph = g
        cur = torch.cuda.current_stream(env.device)
        cur.wait_stream(side)
        env.use_stream(cur)                                     # replays run on the caller's current stream
        env.step_count = t0                                     # host-side counter advanced while capturing
        self.obs[0].copy_(state)

    def collect(self):
        """Run one rollout; buffers are valid after the current stream's work completes."""
        if self._graph is not None:
            self._graph.replay()
            self.env.step_count = self.env.step_count + self.T   # the captured launches carry no host-side effects
        else:
            self._body()
        return self

    def roll_over(self):
        """Make the last observation the first one of the next rollout."""
        self.obs[0].copy_(self.obs[self.T])


def _rowmap(s: int, h: int, j: int) -> int:
    """Row of the previous layer's output that MFMA B-slot (h, j) of K-slice s carries (csrc/rmav_policy_mfma.hpp)."""
    r = 8 * (s & 1) + j
    return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h


class _PolicyPacker:
    """Packs an ``MlpPolicy`` into the weight buffer ``rmav_rollout_policy`` reads (layouts in include/rmav.h).

    The layouts are fixed permutations (+ zero padding) of the flattened parameters, so the index map is
    built once (NumPy, host) and every repack is: one ``cat`` of the parameters, one gather, and for the
    bf16 fragments one dtype conversion - a handful of launches, cheap enough to run before every rollout."""

    K_TANH = 2.8853900817779268   # 2 log2(e): the f16 actor's layer-2 fragments carry -2 k W2, layer 3's -2 W3 (include/rmav.h)

    def __init__(self, policy: MlpPolicy, n_obs: int, bf16_mfma: bool, f32_mfma: bool = False, f16_mfma: bool = False):
        import numpy as np

        H = policy.pi[0].out_features
        assert H == 64 and policy.pi[1].in_features == 64, "the in-kernel policy is the 2 x 64 baselines mlp"
        assert n_obs <= 16
        # f16: the bf16 fragment layout with f16 pairs (and the tanh fold's weight scales)
        self.f16 = bool(f16_mfma)
        self.shared = bool(getattr(policy, "shared", False))
        assert not self.shared or self.f16, "the shared-trunk policy runs on the f16 actor only (RMAV_POLICY_F16_SHARED)"
        self.policy, self.bf16, self.f32m = policy, bool(bf16_mfma) or self.f16, bool(f32_mfma)
        assert not (self.bf16 and self.f32m)
        if self.shared:
            self._init_shared(policy, n_obs)
            return
        self.params = [policy.pi[0].weight, policy.pi[0].bias, policy.pi[1].weight, policy.pi[1].bias,
                       policy.pi[2].weight, policy.pi[2].bias, policy.vf[0].weight, policy.vf[0].bias,
                       policy.vf[1].weight, policy.vf[1].bias, policy.vf[2].weight, policy.vf[2].bias, policy.logstd]
        offs = np.cumsum([0] + [p.numel() for p in self.params])
        ZERO = int(offs[-1])                       # index of the appended 0.0
        n_act = policy.logstd.numel()

        def W(net, layer, r, c):                   # flat index of weight[r][c] (or ZERO when outside the matrix)
            p = self.params[6 * net + 2 * layer]
            rows, cols = p.shape
            return int(offs[6 * net + 2 * layer]) + r * cols + c if (r < rows and c < cols) else ZERO

        def Bv(net, layer, r):
            p = self.params[6 * net + 2 * layer + 1]
            return int(offs[6 * net + 2 * layer + 1]) + r if r < p.numel() else ZERO

        logstd = [int(offs[12]) + c if c < n_act else ZERO for c in range(4)]
        if self.f32m:   # A operands of v_mfma_f32_32x32x2_f32 (include/rmav.h, csrc/rmav_policy_mfma32.hpp)
            def row(r, h):
                return (r & 3) + 8 * (r >> 2) + 4 * h

            idx = []
            for net in range(2):
                for T in range(2):               # A1[T][sq][lane][j] = W1p[32 T + m][2 (4 sq + j) + h]
                    for sq in range(2):
                        idx += [W(net, 0, 32 * T + (l & 31), 2 * (4 * sq + jj) + (l >> 5)) for l in range(64) for jj in range(4)]
                for To in range(2):              # A2[To][Tin][rq][lane][j] = W2[32 To + m][32 Tin + row(4 rq + j, h)]
                    for Tin in range(2):
                        for rq in range(4):
                            idx += [W(net, 1, 32 * To + (l & 31), 32 * Tin + row(4 * rq + jj, l >> 5)) for l in range(64) for jj in range(4)]
                for h in range(2):               # W3[h][o][16 Tin + r] = W3p[o][32 Tin + row(r, h)]
                    for o in range(4):
                        idx += [W(net, 2, o, 32 * Tin + row(r, h)) for Tin in range(2) for r in range(16)]
                idx += [Bv(net, 0, jj) for jj in range(64)] + [Bv(net, 1, jj) for jj in range(64)] + [Bv(net, 2, k) for k in range(4)]
            idx += logstd
            self.idx_f32 = torch.tensor(idx, dtype=torch.int64, device=policy.logstd.device)
            self.n_out = len(idx)
        elif not self.bf16:
            nsp = (n_obs + 3) // 4 * 4
            idx = []
            for net in range(2):
                idx += [W(net, 0, jj, i) for jj in range(H) for i in range(nsp)]          # W1 [H][NSP]
                idx += [Bv(net, 0, jj) for jj in range(H)]
                idx += [W(net, 1, jj, i) for i in range(H) for jj in range(H)]            # W2T[i][j] = W2[j][i]
                idx += [Bv(net, 1, jj) for jj in range(H)]
                idx += [W(net, 2, k, jj) for jj in range(H) for k in range(4)]            # W3T[j][k] = W3[k][j]
                idx += [Bv(net, 2, k) for k in range(4)]
            idx += logstd
            self.idx_f32 = torch.tensor(idx, dtype=torch.int64, device=policy.logstd.device)
            self.n_out = len(idx)
        else:
            frag, f32 = [], []                     # per net: bf16 fragment indices, fp32 (bias) indices
            for net in range(2):
                fr = []
                for Mt in range(2):                # A1[Mt][lane][j] = W1p[32 Mt + m][8 h + j]
                    fr += [W(net, 0, 32 * Mt + (l & 31), 8 * (l >> 5) + jj) for l in range(64) for jj in range(8)]
                for Mt in range(2):                # A2[Mt][s][lane][j] = W2[32 Mt + m][rowmap(s, h, j)]
                    for s_ in range(4):
                        fr += [W(net, 1, 32 * Mt + (l & 31), _rowmap(s_, l >> 5, jj)) for l in range(64) for jj in range(8)]
                for s_ in range(4):                # A3[s][lane][j] = W3p[m][rowmap(s, h, j)]
                    fr += [W(net, 2, l & 31, _rowmap(s_, l >> 5, jj)) for l in range(64) for jj in range(8)]
                frag.append(fr)
                f32.append([Bv(net, 0, r) for r in range(64)] + [Bv(net, 1, r) for r in range(64)] +
                           [Bv(net, 2, r) for r in range(32)])
            dev = policy.logstd.device
            self.idx_frag = torch.tensor(frag[0] + frag[1], dtype=torch.int64, device=dev)
            self.idx_bias = torch.tensor(f32[0] + f32[1] + logstd, dtype=torch.int64, device=dev)
            self.n_frag = len(frag[0]) // 2        # floats per net of fragments (2 bf16 per float)
            self.n_bias = len(f32[0])
            self.n_out = 2 * (self.n_frag + self.n_bias) + 4
            if self.f16:   # per fragment element: 1 (layer 1), -2k (layer 2), -2 (layer 3)
                per_net = [1.0] * (2 * 64 * 8) + [-2.0 * self.K_TANH] * (8 * 64 * 8) + [-2.0] * (4 * 64 * 8)
                self.frag_scale = torch.tensor(per_net + per_net, dtype=torch.float32, device=dev)

    def _init_shared(self, policy, n_obs):
        """ONE net in the bf16 fragment layout (include/rmav.h, RMAV_POLICY_F16_SHARED): output rows 0..3 of the padded 32-row output
        tile = the mean head, row 4 = the value head; then logstd [4]."""
        import numpy as np

        self.params = [policy.pi[0].weight, policy.pi[0].bias, policy.pi[1].weight, policy.pi[1].bias, policy.pi[2].weight,
                       policy.pi[2].bias, policy.vf[0].weight, policy.vf[0].bias, policy.logstd]
        offs = np.cumsum([0] + [p.numel() for p in self.params])
        ZERO, n_act = int(offs[-1]), policy.logstd.numel()
        assert n_act <= 4

        def W(k, r, c):
            rows, cols = self.params[k].shape
            return int(offs[k]) + r * cols + c if (r < rows and c < cols) else ZERO

        def W3(m, c):      # padded output layer: rows 0 .. nA-1 the mean head, row 4 the value head
            return W(4, m, c) if m < n_act else (W(6, 0, c) if m == 4 else ZERO)

        fr = []
        for Mt in range(2):
            fr += [W(0, 32 * Mt + (l & 31), 8 * (l >> 5) + jj) for l in range(64) for jj in range(8)]
        for Mt in range(2):
            for s_ in range(4):
                fr += [W(2, 32 * Mt + (l & 31), _rowmap(s_, l >> 5, jj)) for l in range(64) for jj in range(8)]
        for s_ in range(4):
            fr += [W3(l & 31, _rowmap(s_, l >> 5, jj)) for l in range(64) for jj in range(8)]
        b3 = [int(offs[5]) + r if r < n_act else (int(offs[7]) if r == 4 else ZERO) for r in range(32)]
        bias = [int(offs[1]) + r for r in range(64)] + [int(offs[3]) + r for r in range(64)] + b3
        logstd = [int(offs[8]) + c if c < n_act else ZERO for c in range(4)]
        dev = policy.logstd.device
        self.idx_frag = torch.tensor(fr, dtype=torch.int64, device=dev)
        self.idx_bias = torch.tensor(bias + logstd, dtype=torch.int64, device=dev)
        self.n_frag, self.n_bias = len(fr) // 2, len(bias)
        self.n_out = self.n_frag + self.n_bias + 4
        per_net = [1.0] * (2 * 64 * 8) + [-2.0 * self.K_TANH] * (8 * 64 * 8) + [-2.0] * (4 * 64 * 8)
        self.frag_scale = torch.tensor(per_net, dtype=torch.float32, device=dev)

    def native_maps(self):
        """(idx_lo, idx_hi) int32 device tensors of ``rmav_pack_policy``: output word i = flat[idx_lo[i]] (idx_hi[i] < 0) or the
        bf16 pair (flat[idx_lo[i]], flat[idx_hi[i]]); built once from the same index lists ``pack`` uses."""
        if getattr(self, "_native", None) is None:
            dev = self.params[0].device
            if self.shared:
                fr, bi = self.idx_frag.to(torch.int32), self.idx_bias.to(torch.int32)
                lo = torch.cat([fr[0::2], bi])
                hi = torch.cat([fr[1::2], torch.full((bi.numel(),), -1, dtype=torch.int32, device=dev)])
            elif self.f32m or not self.bf16:
                lo = self.idx_f32.to(torch.int32)
                hi = torch.full_like(lo, -1)
            else:
                fr, bi, nf, nb = self.idx_frag.to(torch.int32), self.idx_bias.to(torch.int32), self.n_frag, self.n_bias
                neg = lambda k: torch.full((k,), -1, dtype=torch.int32, device=dev)  # noqa: E731
                # pack(): [fragments net 0 | biases net 0 | fragments net 1 | biases net 1 | logstd]; a fragment word = bf16 pair
                lo = torch.cat([fr[0:2 * nf:2], bi[:nb], fr[2 * nf::2], bi[nb:2 * nb], bi[2 * nb:]])
                hi = torch.cat([fr[1:2 * nf:2], neg(nb), fr[2 * nf + 1::2], neg(nb), neg(bi.numel() - 2 * nb)])
            assert lo.numel() == self.n_out == hi.numel()
            self._native = (lo.contiguous(), hi.contiguous())
        return self._native

    def pack_native(self, env, out: torch.Tensor) -> torch.Tensor:
        """The same buffer as ``pack`` in ONE launch on the env's stream (``rmav_pack_policy``)."""
        import ctypes as C

        from . import _abi as A

        lo, hi = self.native_maps()
        ps = [p.detach() for p in self.params]
        assert all(p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda for p in ps), "parameters must be contiguous fp32 CUDA tensors"
        n = len(ps)
        ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in ps])
        sizes = (C.c_int64 * n)(*[p.numel() for p in ps])
        fn = A.lib().rmav_pack_policy_f16 if self.f16 else A.lib().rmav_pack_policy
        A.check(fn(env._h, n, ptrs, sizes, C.c_void_p(lo.data_ptr()), C.c_void_p(hi.data_ptr()), self.n_out, C.c_void_p(out.data_ptr())))
        return out

    def pack(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in self.params] + [torch.zeros(1, device=self.params[0].device)])
            if self.shared:
                fr = (flat[self.idx_frag] * self.frag_scale).to(torch.float16).view(torch.int16).view(torch.float32)
                res = torch.cat([fr, flat[self.idx_bias]])
            elif self.f32m or not self.bf16:
                res = flat[self.idx_f32]
            else:
                if self.f16:
                    fr = (flat[self.idx_frag] * self.frag_scale).to(torch.float16).view(torch.int16).view(torch.float32)
                else:
                    fr = flat[self.idx_frag].to(torch.bfloat16).view(torch.int16).view(torch.float32)   # [2 * n_frag]
                bi = flat[self.idx_bias]
                res = torch.cat([fr[:self.n_frag], bi[:self.n_bias], fr[self.n_frag:], bi[self.n_bias:2 * self.n_bias],
                                 bi[2 * self.n_bias:]])
            if out is not None:
                out.copy_(res)
                return out
            return res.contiguous()


def pack_policy_weights(policy: MlpPolicy, n_obs: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 layout: per net  W1 [64][NSP] | b1 | W2^T [64][64] | b2 | W3^T [64][4] | b3 [4],  then logstd [4]."""
    return _PolicyPacker(policy, n_obs, False).pack(out)


def pack_policy_weights_f32_mfma(policy: MlpPolicy, n_obs: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 A operands of v_mfma_f32_32x32x2_f32: per net A1 [2][2][64][4] | A2 [2][2][4][64][4] | W3 [2][4][32] |
    b1 [64] | b2 [64] | b3 [4], then logstd [4] (include/rmav.h)."""
    return _PolicyPacker(policy, n_obs, False, f32_mfma=True).pack(out)


def pack_policy_weights_bf16(policy: MlpPolicy, n_obs: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 MFMA fragments: per net A1 [2][64][8] | A2 [2][4][64][8] | A3 [4][64][8] (bf16) | b1 [64] | b2 [64] |
    b3 [32] (fp32), then logstd [4]."""
    return _PolicyPacker(policy, n_obs, True).pack(out)


def pack_policy_weights_f16(policy: MlpPolicy, n_obs: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f16 MFMA fragments in the bf16 layout, layers 2 / 3 pre-scaled by -2 * 2 log2(e) / -2 (tanh folded into the next
    layer: the kernel hands (1 - tanh z) / 2 on and derives the matching biases from these rounded weights)."""
    return _PolicyPacker(policy, n_obs, False, f16_mfma=True).pack(out)


class FusedPolicyCollector:
    """Same buffers and semantics as :class:`RolloutCollector`, but ONE kernel launch per rollout: the policy
    (2 x 64 tanh MLP + value net, weights staged in LDS) is evaluated inside the rollout kernel by the lane
    that owns the env (``rmav_rollout_policy``), so nothing but the trajectory touches HBM."""

    def __init__(self, env: BatchedQuadrotor, policy: MlpPolicy, nsteps: int, bf16_mfma: bool = False,
                 f32_mfma: Optional[bool] = None, native_pack: bool = True, f16_mfma: bool = False):
        """Actor arithmetic: fp32 on the fp32-input matrix instructions (``v_mfma_f32_32x32x2_f32``; the default),
        ``f32_mfma=False`` fp32 FMAs on the vector ALU (same precision class - only the summation order differs - at
        half the speed), ``bf16_mfma=True`` bf16 operands on the matrix cores (~1e-2 on means), ``f16_mfma=True`` f16
        operands with tanh folded into the next layer (the fastest, ~1e-3 on means; csrc/rmav_policy_pair.hpp)."""
        import ctypes as C

        assert not (bf16_mfma and f16_mfma)
        shared = bool(getattr(policy, "shared", False))
        if shared:   # one trunk, two heads: RMAV_POLICY_F16_SHARED (the only actor of that architecture)
            assert not bf16_mfma and not f32_mfma, "a shared-trunk policy runs on the f16 actor"
            f16_mfma = True
        if f32_mfma is None:
            f32_mfma = not (bf16_mfma or f16_mfma)

        from . import _abi as A

        assert env.auto_reset, "rollouts need VecEnv semantics (auto-reset)"
        self.env, self.policy, self.T = env, policy, int(nsteps)
        self.bf16_mfma, self.f32_mfma, self.f16_mfma = bool(bf16_mfma), bool(f32_mfma), bool(f16_mfma)
        self._C, self._A = C, A
        dev = torch.device("cuda", env.device)
        N, nS, nA, T = env.num_envs, env.nS, env.nA, self.T
        f32 = dict(dtype=torch.float32, device=dev)
        self.obs = torch.empty((T + 1, nS, N), **f32)
        self.act = torch.empty((T, nA, N), **f32)
        self.logp = torch.empty((T, N), **f32)
        self.val = torch.empty((T + 1, N), **f32)
        self.rew = torch.empty((T, N), **f32)
        self.done = torch.empty((T, N), dtype=torch.uint8, device=dev)
        n_w = (A.lib().rmav_policy_weight_count_shared() if shared else
               A.lib().rmav_policy_weight_count_bf16() if (self.bf16_mfma or self.f16_mfma) else
               A.lib().rmav_policy_weight_count_f32_mfma() if self.f32_mfma else A.lib().rmav_policy_weight_count(env.kind))
        self.weights = torch.empty(n_w, **f32)
        assert self.weights.data_ptr() % 16 == 0
        self._packer = _PolicyPacker(policy, env.nS, self.bf16_mfma, f32_mfma=self.f32_mfma, f16_mfma=self.f16_mfma)
        assert self._packer.n_out == n_w, (self._packer.n_out, n_w)
        self.obs[0].copy_(env.get_state(layout="soa", device_out=True))
        # The weight repack before every rollout: one gather launch behind the C ABI (rmav_pack_policy).  As ~8 dependent torch
        # launches (cat, gather, bf16 conversion, cat, copy) it cost ~38 us of a 0.21 ms rollout at 65 536 envs x 32 steps; a
        # hipGraph of those launches replayed no faster (the dependent-launch floor, not the host, is what they cost).
        self.native_pack = bool(native_pack)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        # (the env's handle is read at call time: env.close() clears it, and a cached copy would hand a freed handle to the library)
        self._call = (A.lib().rmav_rollout_policy, self.T, p(self.weights), p(self.act), p(self.obs[1:]), p(self.rew), p(self.done),
                      p(self.logp), p(self.val),
                      A.POLICY_F16_SHARED if shared else A.POLICY_F16_MFMA if self.f16_mfma else A.POLICY_BF16_MFMA if self.bf16_mfma
                      else A.POLICY_FP32_MFMA if self.f32_mfma else A.POLICY_FP32)

    def _pack(self):
        if self.native_pack:
            self._packer.pack_native(self.env, self.weights)
        else:
            self._packer.pack(out=self.weights)

    def collect(self):
        if self.env._h is None:
            raise self._A.RmavError(self._A.ERR_INVALID, "the env of this collector is closed")
        self._pack()
        fn = self._call[0]
        self._A.check(fn(self.env._h, *self._call[1:]))
        return self

    def roll_over(self):
        self.obs[0].copy_(self.obs[self.T])


def gae(rew, val, done, gamma: float = 0.99, lam: float = 0.95):
    """Generalised advantage estimation on time-major tensors - the plain torch fp32 form (T small launches per
    call), kept as the reference the HIP kernel (``BatchedQuadrotor.gae`` -> ``rmav_gae``) is tested against and
    for CPU tensors; ``PPO.update`` uses the kernel.

    rew [T,N], val [T+1,N] (val[T] = bootstrap value), done [T,N] (1 = the episode ended with step t; the
    next obs is a fresh reset and must not be bootstrapped from).  Returns (adv [T,N], returns [T,N])."""
    T = rew.shape[0]
    adv = torch.empty_like(rew)
    last = torch.zeros_like(rew[0])
    nonterm = 1.0 - done.to(rew.dtype)
    for t in range(T - 1, -1, -1):
        delta = rew[t] + gamma * val[t + 1] * nonterm[t] - val[t]
        last = delta + gamma * lam * nonterm[t] * last
        adv[t] = last
    return adv, adv + val[:T]


class PPO:
    """Clipped-surrogate PPO (baselines ppo2 defaults: lr 3e-4, clip 0.2, 4 epochs x 4 minibatches,
    vf_coef 0.5, ent_coef 0, max_grad_norm 0.5, gamma 0.99, lambda 0.95)."""

    def __init__(self, policy: MlpPolicy, lr: float = 3e-4, clip: float = 0.2, epochs: int = 4, minibatches: int = 4,
                 vf_coef: float = 0.5, ent_coef: float = 0.0, max_grad_norm: float = 0.5, gamma: float = 0.99,
                 lam: float = 0.95, reward_scale: float = 1.0):
        self.policy = policy
        self.reward_scale = float(reward_scale)   # baselines' --reward_scale (gym_reinmav/run.py:76)
        self.opt = torch.optim.Adam(policy.parameters(), lr=lr, eps=1e-5)
        self.clip, self.epochs, self.minibatches = clip, epochs, minibatches
        self.vf_coef, self.ent_coef, self.max_grad_norm, self.gamma, self.lam = vf_coef, ent_coef, max_grad_norm, gamma, lam

    def _allreduce_grads(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        params = [p for p in self.policy.parameters() if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
        o = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n

    def loss(self, obs, act, logp_old, val_old, adv, ret):
        """Clipped-surrogate loss of one minibatch (feature-major obs [nS, B], act [nA, B]; the rest [B]);
        the advantages are normalised over the minibatch as in baselines' ppo2.  -> (loss, pg, vf, ratio)."""
        a = (adv - adv.mean()) / (adv.std() + 1e-8)
        mean, v = self.policy(obs)
        logp = self.policy.log_prob(mean, act)
        ratio = torch.exp(logp - logp_old)
        pg = torch.max(-a * ratio, -a * torch.clamp(ratio, 1 - self.clip, 1 + self.clip)).mean()
        v_clip = val_old + torch.clamp(v - val_old, -self.clip, self.clip)
        vf = 0.5 * torch.max((v - ret) ** 2, (v_clip - ret) ** 2).mean()
        return pg + self.vf_coef * vf - self.ent_coef * self.policy.entropy(), pg, vf, ratio

    def update(self, ro: RolloutCollector) -> dict:
        T, N = ro.rew.shape
        if ro.rew.is_cuda:   # one HIP launch over the [T][N] trajectory (per-lane reverse scan, csrc/rmav_gae.hpp)
            adv, ret = ro.env.gae(ro.rew, ro.done, ro.val, self.gamma, self.lam, self.reward_scale)
        else:
            rew = ro.rew if self.reward_scale == 1.0 else ro.rew * self.reward_scale
            adv, ret = gae(rew, ro.val, ro.done, self.gamma, self.lam)
        obs = ro.obs[:T].permute(1, 0, 2).reshape(ro.obs.shape[1], T * N)   # [nS, T*N] feature-major
        act = ro.act.permute(1, 0, 2).reshape(ro.act.shape[1], T * N)
        logp_old, val_old = ro.logp.reshape(-1), ro.val[:T].reshape(-1)
        adv, ret = adv.reshape(-1), ret.reshape(-1)
        stats = {}
        B = T * N
        mb = B // self.minibatches
        for _ in range(self.epochs):
            perm = torch.randperm(B, device=obs.device)
            for i in range(self.minibatches):
                idx = perm[i * mb:(i + 1) * mb]
                loss, pg, vf, ratio = self.loss(obs[:, idx], act[:, idx], logp_old[idx], val_old[idx], adv[idx], ret[idx])
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                self._allreduce_grads()
                torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
                self.opt.step()
                stats = {"pg_loss": pg.detach(), "vf_loss": vf.detach(), "ratio_max": ratio.detach().max()}
        var_y = ret.var()
        stats["explained_variance"] = 1.0 - (ret - val_old).var() / (var_y + 1e-8)
        return {k: float(v) for k, v in stats.items()}


def sync_parameters(policy: MlpPolicy, src: int = 0):
    """Broadcast rank ``src``'s parameters (data-parallel start)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for p in policy.parameters():
            dist.broadcast(p.data, src)
