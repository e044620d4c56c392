#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/profile_actors.sh into a per-actor table (markdown on stdout) and
<dir>/actor_instr_mix.json (instructions per 64 envs and env-step by class, as the SQ counters measured them), which
bench.py's policy leg uses for its vector-pipe roofline."""
import collections, csv, glob, json, os, re, sys

d = sys.argv[1]
N, T = int(os.environ.get("N", "65536")), int(os.environ.get("T", "32"))
ACTORS = collections.OrderedDict([   # kernel-name pattern -> label (quadrotor3d = kind 2)
    (r"k_rollout<2, 3, 0,", "fp32_valu"), (r"k_rollout<2, 8, 0,", "fp32_mfma"), (r"k_rollout<2, 4, 0,", "bf16_1w"),
    (r"k_rollout_pair<2, 0>", "bf16_mfma"), (r"k_rollout_pair<2, 1>", "f16_mfma"), (r"k_rollout_pair_shared<2>", "f16_shared")])


def label(kname):
    for pat, lab in ACTORS.items():
        if pat in kname:
            return lab
    return None


dur = collections.defaultdict(list)
for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        lab = label(r["Kernel_Name"])
        if lab:
            dur[lab].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        lab = label(r["Kernel_Name"])
        if lab:
            cnt[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / max(1, len(v))
# NOTE tools/actor_bench.py launches the pair kernels with 1, 2 and 4 pairs per workgroup: all of them are averaged here
# (their instruction counts are identical; durations differ by a few %).
units = N / 64 * T     # (64 envs) x env-steps per launch
mix = {}
print(f"# SQ counters of the policy-in-kernel rollouts (quadrotor3d, {N} envs x {T} steps; per-dispatch means)\n")
# Vector-pipe model (tools/micro/issue_rate.hip, profiles/r04/issue_rate.md): cycles one wave64 instruction occupies the SIMD's vector
# pipe with two or more wavefronts on it - transcendental 8.4, conversion / packed 4.45 - 5.0, anything else 2.8; a LONE wavefront
# cannot issue faster than one instruction per 5.3 cycles (transcendental 8.8).  PACKED: static count of v_pk_* f32 per 64 envs·step.
PACKED = {"fp32_valu": 0, "fp32_mfma": 0, "bf16_1w": 236, "bf16_mfma": 252, "f16_mfma": 130, "f16_shared": 64}
print("| actor | kernel us (trace) | per 64 envs·step: VALU | transcendental | convert | MFMA | SALU | LDS | VMEM wr | SIMD cycles per 64 envs·step | "
      "wavefront: issuing % | s_waitcnt / barrier % | issue-stall % | VALU busy per SIMD % | MFMA busy per SIMD % | vector-pipe model cycles | model / measured |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for lab in ACTORS.values():
    c = {k: mean(v) for k, v in cnt[lab].items()}
    if not c or not dur[lab]:
        continue
    g = lambda k: c.get(k, float("nan"))
    waves = g("SQ_WAVES")
    per = lambda k: g(k) / units
    wc = g("SQ_WAVE_CYCLES")
    busy = g("SQ_BUSY_CYCLES")
    row = dict(valu=per("SQ_INSTS_VALU"), trans=per("SQ_INSTS_VALU_TRANS_F32"), cvt=per("SQ_INSTS_VALU_CVT"), mfma=per("SQ_INSTS_MFMA"),
               salu=per("SQ_INSTS_SALU"), lds=per("SQ_INSTS_LDS"), vmem_wr=per("SQ_INSTS_VMEM_WR"), waves_per_64_envs=waves / (N / 64),
               kernel_us=mean(dur[lab]))
    kcyc = g("GRBM_GUI_ACTIVE") / 8                      # kernel duration in shader cycles (the counter sums the 8 XCDs)
    simd_cyc = kcyc * 1024 / units                        # SIMD-cycles available per 64 envs and env-step (256 CUs x 4 SIMDs)
    lone = row["waves_per_64_envs"] * (N / 64) <= 1024    # one wavefront per SIMD: issue-limited
    other = row["valu"] - row["trans"] - row["cvt"] - row["mfma"] - PACKED[lab]
    model = ((row["trans"] * 8.8 + (row["valu"] - row["trans"] - row["mfma"]) * 5.3) if lone else
             (row["trans"] * 8.4 + row["cvt"] * 4.45 + PACKED[lab] * 5.0 + other * 2.8))
    row.update(clock_ghz=kcyc / (mean(dur[lab]) * 1e3), simd_cycles=simd_cyc, packed_static=PACKED[lab], model_cycles=model, model_frac=model / simd_cyc, lone_wavefront=bool(lone))
    mix[lab] = row
    print(f"| {lab} | {mean(dur[lab]):.1f} | {row['valu']:.0f} | {row['trans']:.0f} | {row['cvt']:.0f} | {row['mfma']:.0f} | {row['salu']:.0f} | {row['lds']:.0f} | "
          f"{row['vmem_wr']:.1f} | {simd_cyc:.0f} | {100 * g('SQ_ACTIVE_INST_ANY') / wc:.0f} | {100 * g('SQ_WAIT_ANY') / wc:.0f} | "
          f"{100 * g('SQ_WAIT_INST_ANY') / wc:.0f} | {100 * g('SQ_ACTIVE_INST_VALU') * 4 / (kcyc * 1024):.0f} | "
          f"{100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (kcyc * 1024):.0f} | {model:.0f} ({'lone-wavefront issue' if lone else 'shared pipe'}) | {model / simd_cyc:.2f} |")
print("\nraw counter means per actor:\n")
for lab in ACTORS.values():
    if cnt[lab]:
        print(f"- **{lab}**: " + ", ".join(f"{k} {mean(v):.4g}" for k, v in sorted(cnt[lab].items())))
json.dump(mix, open(os.path.join(d, "actor_instr_mix.json"), "w"), indent=1)
