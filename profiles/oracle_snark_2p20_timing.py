#!/usr/bin/env python3
"""Times the ORACLE's SNARK::prove at the metric's own size (2^20 constraints, bench.py's instance and tape) with 1 thread
and with all cores, so that bench.py's 2^17-sample `cpu_baseline` (constraints/s is flat in the size) is backed by a
measurement at 2^20. Writes profiles/r3_oracle_snark_2p20_timing.json. Test infrastructure only (oracle/).
Run from the repo root:  python profiles/oracle_snark_2p20_timing.py [log2_size]"""
import ctypes, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import load_oracle, sz, vp, u64x4

s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 1 << s
orc = load_oracle()
cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
res = {"workload": f"SNARK::prove, produce_synthetic_r1cs(2^{s}, 2^{s}, 10), seed 0, tape seed 100", "cpu_model": cpu[0] if cpu else "?",
       "logical_cores": os.cpu_count(), "runs": []}
t0 = time.time()
inst = vp(orc.orc_instance_synthetic(sz(N), sz(N), sz(10), ctypes.c_uint64(0)))
g = vp(orc.orc_snark_gens_new(sz(N), sz(N), sz(10), sz(N)))
orc.orc_set_threads(ctypes.c_int(os.cpu_count() or 1))
e = vp(orc.orc_snark_encode(inst, g))
res["setup_seconds"] = round(time.time() - t0, 1)
tape = u64x4(); orc.orc_seed_scalar(b"tape", ctypes.c_uint64(100), tape)
for threads in (os.cpu_count() or 1, 1):
    orc.orc_set_threads(ctypes.c_int(threads))
    times = (ctypes.c_double * 10)()
    t0 = time.time()
    p = vp(orc.orc_snark_prove(inst, g, e, b"snark_example", tape, times))
    dt = time.time() - t0
    n = orc.orc_proof_bytes(p, None, sz(0)); b = (ctypes.c_uint8 * n)(); orc.orc_proof_bytes(p, b, sz(n))
    res["runs"].append({"threads": threads, "seconds": round(dt, 2), "constraints_per_s": round(N / dt, 1),
                        "proof_sha256": hashlib.sha256(bytes(b)).hexdigest(),
                        "phases_s": dict(zip(["polycommit", "sc_phase_one", "sc_phase_two", "polyeval", "r1cs_sat", "eval_sparse_polys",
                                              "commit_nondet_witness", "build_layered_network", "evalproof_layered_network", "total"],
                                             [round(x, 3) for x in times]))})
    orc.orc_proof_free(p)
    print(res["runs"][-1], flush=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", f"r3_oracle_snark_2p{s}_timing.json"), "w"), indent=1)
