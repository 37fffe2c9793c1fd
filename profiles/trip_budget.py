#!/usr/bin/env python3
"""The trip budget of one SNARK::prove: which C-ABI entry points the proving thread blocks in, per proof, and what the measured
options for removing trips are worth. Inputs are the files of one collection (profiles/collect_r6.sh + collect_r6_diag.sh):
  callstats.txt (option host.callstats: wall time per entry point), ktime_probe.txt (in-kernel stamps of the two latency kernels),
  trip_probe.txt (bench/trip_probe: a trip beyond its kernel), ab_launch_ahead.txt (interleaved A/B of option sumcheck.launch_ahead).
usage: python profiles/trip_budget.py <dir with those files> <ms per proof> <trips per proof>"""
import re, sys, os
d, ms_proof, trips = sys.argv[1], float(sys.argv[2]), int(float(sys.argv[3]))
rows = []
seen = set()
for line in open(os.path.join(d, "callstats.txt")):
    m = re.match(r"\[callstats\]\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
    if m and m.group(1) not in seen:   # the file holds the last proofs of the run back to back: one table is enough
        seen.add(m.group(1)); rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
rows.sort(key=lambda r: -r[2])
groups = [
    ("waiting for a commitment's MSM (sp_job_wait: the part of the two big commitments nothing could be overlapped with)", ("sp_job_wait",)),
    ("batched cubic sum-check, two-rounds-per-trip kernels (k_cubic_bind2_eval)", ("sp_sumcheck_bind2_eval_batched", "sp_sumcheck_bind2_eval_tables_batched", "sp_sumcheck_eval_coeffs_batched")),
    ("batched cubic sum-check, throughput-sized rounds", ("sp_sumcheck_bind_eval_batched_eq", "sp_sumcheck_eval_batched_eq", "sp_sumcheck_bind_eval_batched", "sp_sumcheck_eval_batched")),
    ("inner-product argument (45 rounds + set-up + last round)", ("sp_ipa_round_lr", "sp_ipa_begin_dev", "sp_ipa_finish_commit", "sp_ipa_round_fold", "sp_ipa_set_scale")),
    ("ZK sum-checks of R1CSProof (the device leg runs under the round's commitments: start .. collect)", ("sp_sumcheck_bind_eval_collect", "sp_sumcheck_bind_eval_start", "sp_sumcheck_eval")),
    ("few-term commitments on the proving core (no trip)", ("sp_host_commit_small", "sp_host_zk_ahead_begin", "sp_host_zk_ahead_wait")),
]
by = {r[0]: r for r in rows}
print("# Round 6: the trip budget of one 2^20 SNARK::prove — %.2f ms per proof, %d completed waits on the main stream (config.fs_trips_per_proof)." % (ms_proof, trips))
print("# Wall time of the proving thread inside the library's entry points (option host.callstats; the thread's own transcript, field and")
print("# small-commitment work between the calls is the rest of the proof).\n")
print("%-118s %6s %9s %8s" % ("what", "calls", "total ms", "avg us"))
acc = 0.0
for title, names in groups:
    n = sum(by[k][1] for k in names if k in by); t = sum(by[k][2] for k in names if k in by)
    acc += t
    print("%-118s %6d %9.3f %8.1f" % (title[:118], n, t, 1e3 * t / n if n else 0.0))
    for k in names:
        if k in by: print("    %-114s %6d %9.3f %8.1f" % (k, by[k][1], by[k][2], by[k][3]))
other = [(r[0], r[1], r[2], r[3]) for r in rows if not any(r[0] in names for _t, names in groups)]
to = sum(r[2] for r in other)
print("%-118s %6d %9.3f" % ("everything else (%d entry points; the five largest below)" % len(other), sum(r[1] for r in other), to))
for r in other[:5]: print("    %-114s %6d %9.3f %8.1f" % r)
print("%-118s %6s %9.3f  = %.0f %% of the proof" % ("inside the library in all", "", acc + to, 100 * (acc + to) / ms_proof))
def grab(name):
    p = os.path.join(d, name)
    return open(p).read() if os.path.exists(p) else ""
tp = grab("trip_probe.txt")
if tp:
    print("\n## a trip beyond its kernel (bench/trip_probe.hip: no library code; A = the product's launch-per-trip, B = launched ahead + bell)")
    print(tp.rstrip())
kt = grab("ktime_probe.txt")
if kt:
    print("\n## inside the two latency kernels (in-kernel stamps, bench/ktime_probe.py; the whole file: r6_ktime_probe.txt)")
    for line in kt.splitlines():
        if line.startswith("====") or "host time of the call" in line or "whole workgroup" in line or "critical path" in line: print(line.rstrip()[:230])
ab = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r6_ab_launch_ahead.txt")).read() if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r6_ab_launch_ahead.txt")) else ""
if ab:
    print("\n## option sumcheck.launch_ahead = 1 against 0, interleaved A/B sessions (profiles/r6_ab_launch_ahead.txt holds all of them)")
    for line in ab.splitlines():
        if "mean" in line or "Verdict" in line or "final protocol" in line or "latency kernels' priority" in line: print(line.lstrip("# ").rstrip()[:200])
# ---- what the options are worth (numbers from the tables above)
def tot(*names): return sum(by[k][2] for k in names if k in by)
def cnt(*names): return sum(by[k][1] for k in names if k in by)
fixed = None
for line in tp.splitlines():
    m = re.match(r"grid 16 x 256, body\s+8 mul-adds: .*A launch-per-trip ([\d.]+) us/trip.*B launched ahead \+ doorbell ([\d.]+)", line)
    if m: fixed = (float(m.group(1)), float(m.group(2)))
print("\n## what removing trips is worth")
if fixed:
    print("* A trip costs %.1f us beyond its kernel's body (launch call, dispatch, completion flag over PCIe, the proving thread's wake-up: trip_probe A with an" % fixed[0])
    print("  empty body); %d trips x %.1f us = %.1f ms of the %.1f ms proof is the ceiling of ANY scheme that keeps the arithmetic where it is." % (trips, fixed[0], trips * fixed[0] / 1e3, ms_proof))
n2 = cnt("sp_sumcheck_bind2_eval_batched", "sp_sumcheck_bind2_eval_tables_batched")
print("* (built, not the default) Launching the next two-rounds kernel ahead of its challenges (sumcheck.launch_ahead = 1; %d of the %d trips qualify: every two-bind" % (n2, trips))
print("  trip whose predecessor is a trip over the same tables): the probe's B against A is 1.3-6 us per trip depending on the body; in the proof the entry")
print("  point's time falls by ~2.5 us per trip when it helps (the bell is one PCIe read away, the decision is relayed to the other workgroups through a")
print("  device word) — 0.3 ms per proof in the first five A/B sessions, nothing in the four later ones: inside the +-0.4 ms run-to-run spread of the proof.")
print("  The same treatment of the %d inner-product rounds and the %d throughput-sized rounds would add (45 + 58) x 2.5 us = 0.26 ms at best: not built." % (cnt("sp_ipa_round_lr"), cnt("sp_sumcheck_bind_eval_batched_eq", "sp_sumcheck_bind_eval_batched", "sp_sumcheck_eval_batched_eq")))
zk = by.get("sp_sumcheck_bind_eval_collect")
if zk:
    print("* Two rounds per trip in the ZK sum-checks: the proving thread's exclusive wait is sp_sumcheck_bind_eval_collect, %d calls, %.2f ms in all = %.1f us per round" % (zk[1], zk[2], zk[3]))
    print("  (the device leg runs under the round's 45-55 us of commitments and transcript work): halving those trips saves at most %.2f ms. Closed." % (zk[2] / 2))
ipa = by.get("sp_ipa_round_lr")
if ipa:
    print("* The %d inner-product rounds: %.2f ms = %.1f us per round against a critical path of 42-52 us inside the kernel (two dependent addition trees of 8 and" % (ipa[1], ipa[2], ipa[3]))
    print("  6-8 levels, each level two field multiplications deep: ktime stamps above). L and R of every round are sums over the ORIGINAL generators, so the last")
    print("  rounds cannot move to the proving core (2 x 2048 terms x 26 additions per round there); folding the generators on the device for a short tail would")
    print("  trade 6 rounds x %.0f us per opening for ~0.5 ms of variable-base arithmetic on the core. Closed with these numbers." % ipa[3])
print("* Fiat-Shamir on the device (transcript, challenges and the round logic in a resident kernel) would remove the trips altogether: at most the %.1f ms of the" % ((trips * fixed[0] / 1e3) if fixed else 0.0))
print("  first bullet. It moves Merlin/Keccak, the round polynomials, the %d few-term commitments (%.1f us each on the" % (cnt("sp_host_commit_small"), by["sp_host_commit_small"][3] if "sp_host_commit_small" in by else 0.0))
print("  core, ~100 us each as device launches, DESIGN 4) and the Sigma-protocols into one kernel's serial thread at a quarter of the core's clock: not built.")
jw = by.get("sp_job_wait")
if jw:
    print("* The largest single item is not a trip at all: sp_job_wait, %.2f ms — the part of the two big commitments' MSM that the chain cannot hide. It is" % jw[2])
    print("  arithmetic at the power-limited rate of the row MSM (DESIGN 3): fewer additions per scalar (the mixed-width tables and the pair planner of this")
    print("  round) is what moved it.")
