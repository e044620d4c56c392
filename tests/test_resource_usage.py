"""Register / scratch budget of the gfx950 kernels, from hipcc's -Rpass-analysis=kernel-resource-usage (`make asm`,
no GPU needed).  A guard against silent codegen regressions: in round 2 an innocent `if (n_steps <= 0) return;` in
front of the step loop cost the one-wavefront rollout kernels 33-38 VGPRs (two to three wavefronts per SIMD of
occupancy) and a `h ? x[a] : x[b]` select put a state array into scratch memory, and nothing failed."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "reinmav-gym_amd")


@pytest.fixture(scope="module")
def usage():
    subprocess.run(["make", "-s", "-C", PKG, "asm"], check=True)
    txt = open(os.path.join(PKG, "build", "resource_usage.txt")).read()
    out = {}
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split(" ")[0]
        out[name] = {k: int(re.search(pat, b).group(1)) for k, pat in (
            ("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
            ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))}
    assert len(out) >= 100
    return out


def _k(usage, kind, mode, st, fixed=0):
    """k_rollout<K, MODE, ST, FIXED> (FIXED: the usual launch options compiled in - two-wavefront kernels, ST_WRITE_THROUGH only)"""
    hits = [v for n, v in usage.items() if n.startswith(f"_ZN4rmav9k_rolloutILi{kind}ELi{mode}ELi{st}ELb{fixed}E")]
    assert len(hits) == 1, (kind, mode, st, fixed)
    return hits[0]


def test_no_kernel_uses_scratch(usage):
    """Everything lives in registers / LDS.  Known exceptions (32 bytes each): the bf16 actor of the 13-state ReinmavEnv, and - since
    gravity is a vector (round 6: two more fp64 constants in the 2-D slung-load step) - the one-wavefront fp32 / bf16 actors of
    quadrotor2d-slungload, kernels that already park 45 values in accumulator registers at one wavefront per SIMD; no BASELINE config
    runs them (C5 is quadrotor3d, and its default actor is the wavefront-pair kernel); the fp32 actor of ReinmavEnv (48 bytes since the
    per-env records of round 6: the byte offset 16 * env is one more short-lived register in a kernel that uses all 256)."""
    known = ("_ZN4rmav9k_rolloutILi4ELi4E", "_ZN4rmav9k_rolloutILi1ELi3ELi0E", "_ZN4rmav9k_rolloutILi1ELi4ELi0E", "_ZN4rmav9k_rolloutILi4ELi3ELi0E")
    bad = {n: v["scratch"] for n, v in usage.items() if v["scratch"] and not n.startswith(known)}
    assert not bad, bad
    assert all(v["scratch"] <= 48 for v in usage.values())


@pytest.mark.parametrize("kind,budget,min_occ", [(0, 64, 7), (1, 104, 4), (2, 80, 6), (3, 144, 3)])
def test_one_wavefront_rollout_register_budget(usage, kind, budget, min_occ):
    """k_rollout<K, ACT_RANDOM | ACT_BUFFER | ACT_CONTROLLER, ST_STREAM | ST_DEFAULT>: the big-batch kernels live on
    occupancy (4-16 wavefronts per SIMD hide the store latency)."""
    for mode, st in ((1, 2), (1, 0), (0, 0)):
        u = _k(usage, kind, mode, st)
        assert u["vgpr"] <= budget and u["occ"] >= min_occ and u["spill"] == 0, (kind, mode, st, u)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_two_wavefront_rollout_fits_1024_threads(usage, kind):
    """The random-action two-wavefront kernels launch up to 8 pairs (1024 threads): <= 128 VGPRs, no spills."""
    for st in (0, 1, 2):
        u = _k(usage, kind, 5, st)
        assert u["vgpr"] <= 128 and u["spill"] == 0, (kind, st, u)
    for mode in (6, 9):            # controller-driven and caller-action variants: 8 pairs as well (round 3)
        u = _k(usage, kind, mode, 1)
        assert u["spill"] == 0 and u["vgpr"] <= 128, (kind, mode, u)
    for mode in (5, 6, 9):         # the same with the usual launch options compiled in (round 4)
        u = _k(usage, kind, mode, 1, fixed=1)
        assert u["spill"] == 0 and u["vgpr"] <= 128, (kind, mode, u)


def test_single_step_kernel_is_small(usage):
    hits = {n: v for n, v in usage.items() if n.startswith("_ZN4rmav6k_stepILi2ELb0ELb0ELi0E")}   # (ST_DEFAULT: the shipped store policy)
    assert len(hits) == 1
    u = next(iter(hits.values()))
    assert u["vgpr"] <= 48 and u["occ"] == 8 and u["lds"] == 0, u


def test_fp32_mfma_actor_leaves_room_for_two_wavefronts_per_simd(usage):
    for kind in (0, 1, 2, 3):
        u = _k(usage, kind, 8, 0)
        assert u["vgpr"] + u["agpr"] <= 256 and u["spill"] == 0, (kind, u)


def test_pair_actors_fit_two_wavefronts_per_simd(usage):
    """k_rollout_pair<K, FMT_BF16 | FMT_F16>: at BASELINE's C5 shape every SIMD hosts two of these wavefronts: <= 256 registers."""
    hits = {n: v for n, v in usage.items() if n.startswith("_ZN4rmav14k_rollout_pairILi") or n.startswith("_ZN4rmav21k_rollout_pair_sharedILi")}
    assert len(hits) == 15
    for n, u in hits.items():
        assert u["vgpr"] + u["agpr"] <= 256 and u["spill"] == 0 and u["scratch"] == 0 and u["occ"] >= 2, (n, u)


def test_matrix_core_kernels_have_no_lds_permutes_and_no_compiler_packed_fp32():
    """gfx950 hazard, root-caused in round 5 (profiles/r05/packed_f32_hazard.md, tools/micro/pk_hazard.hip): a packed-fp32 instruction whose
    op_sel selects the HIGH half of src1 for the low result reads zero in lanes 48..63 while a v_mfma_f32_32x32x16_{f16,bf16} executes on
    the SIMD.  Only the SLP vectoriser emits that form, so NO kernel of the library may contain it (both translation units are built with
    -fno-slp-vectorize, and the Makefile's check_isa step refuses an object that has it).  The matrix-core kernels additionally stay free
    of v_pk_mul_f32 / v_pk_mov_b32 (compiler-only forms) and of LDS permutes (lanes are exchanged with v_permlane32_swap)."""
    subprocess.run(["make", "-s", "-C", PKG, "asm"], check=True)
    bad_form = re.compile(r"v_pk_(fma|mul|add)_f32 .*op_sel:\[[01],1")
    for unit in ("rmav_abi", "rmav_policy_abi"):
        txt = open(os.path.join(PKG, "build", unit + ".gfx950.s")).read()
        hits = bad_form.findall(txt)
        assert not hits, (unit, len(hits))
    txt = open(os.path.join(PKG, "build", "rmav_policy_abi.gfx950.s")).read()
    bodies = re.split(r"^(_ZN4rmav\w+):[^\n]*\n", txt, flags=re.M)   # [pre, name, body, name, body, ...]
    seen = 0
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = body.split(".Lfunc_end")[0]
        if "v_mfma" not in body and not re.match(r"_ZN4rmav9k_rolloutILi\dELi[48]E", name):
            continue                                  # (the fp32 vector-ALU actor: no matrix instructions)
        seen += 1
        for bad in ("ds_bpermute", "ds_permute", "v_pk_mul_f32", "v_pk_mov_b32"):
            assert bad not in body, (name, bad)
    assert seen >= 26, seen   # 10 + 5 pair kernels, 5 + 5 one-wavefront bf16 / fp32-MFMA kernels, mlp_mfma
    assert txt.count("v_permlane32_swap") >= 100 and txt.count("v_mfma_f32_32x32x16_f16") >= 100


def test_makefile_refuses_an_object_with_the_hazardous_instruction_form(tmp_path):
    """The check_isa step: building the policy translation unit WITH the SLP vectoriser must fail, not produce a library."""
    r = subprocess.run(["make", "-C", PKG, "build/rmav_policy_abi.o", "-B", "HIPFLAGS=--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC"],
                       capture_output=True, text=True)
    try:
        assert r.returncode != 0 and "refusing to build" in r.stdout + r.stderr, (r.stdout[-500:], r.stderr[-500:])
        assert not os.path.exists(os.path.join(PKG, "build", "rmav_policy_abi.o"))
    finally:   # put the real object back (the library itself was not relinked: only the one target was requested)
        subprocess.run(["make", "-s", "-C", PKG], check=True)


def _kernel_asm(prefix, which="rmav_abi"):
    path = os.path.join(PKG, "build", which + ".gfx950.s")
    out, on = [], False
    for line in open(path):
        if not on and line.startswith(prefix):
            on = True
        if on:
            out.append(line)
            if "s_endpgm" in line:
                break
    assert out, prefix
    return out


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", [5, 6])
def test_two_wavefront_kernels_have_no_waterfall_loops(usage, kind, mode):
    """hipcc wraps a buffer access in a v_readfirstlane / s_and_saveexec loop when it cannot prove the descriptor
    wave-uniform (guide T20).  The memory wavefront issues ~30 stores per env-step: in round 2 a harmless-looking
    rewrite of one bounds expression made every one of them a loop (243 v_readfirstlane in the quadrotor3d kernel)."""
    asm = _kernel_asm(f"_ZN4rmav9k_rolloutILi{kind}ELi{mode}ELi1E")
    n_rfl = sum("v_readfirstlane" in l for l in asm)
    assert n_rfl <= 8, n_rfl
