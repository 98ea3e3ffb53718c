#!/usr/bin/env python3
"""Turn the raw output of scripts/profile_round.sh into the small files that go under profiles/:
  <tag>_bench_n1.json            the default bench line
  <tag>_kernel_stats.csv         rocprofv3 --stats rows of the library's kernels (name, calls, total / average / min / max ns)
  <tag>_pmc_summary.txt          HBM counters of the headline kernel + traffic.json
  <tag>_mc_pmc.json, mc_flop.json   Mohr-Coulomb counters; flop per plastic point for bench.py's secondary roofline
  <tag>_device_loop_pmc.json, consumer_flop.json   consumer-side kernels: HBM bytes, SQ ratios, issued fp64 work (the bench reads the latter)
  <tag>_icnn_pmc.json, <tag>_field_pmc.json
usage: summarize_round.py gpurun_out/<tag> <tag>"""
import collections
import csv
import json
import pathlib
import subprocess
import sys

out, tag = pathlib.Path(sys.argv[1]), sys.argv[2]
ours = ("vm_", "mc_", "icnn_", "isihara", "heat_", "operand_", "adjoint", "tangent_", "node_sum", "assign", "stream_probe", "cond_", "arena_sweep")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")


# ---- kernel stats
rows = []
for f in (out / "stats").rglob("*kernel_stats.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = short(r["Name"])
            if n.startswith(ours):
                rows.append((n[:100], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
rows.sort(key=lambda r: -r[2])
if rows:
    with open(out / f"{tag}_kernel_stats.csv", "w") as fh:
        fh.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
        for r in rows:
            fh.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.1f},{r[4]:.0f},{r[5]:.0f}\n")
    print("== kernel stats (rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu --no-probe --no-e2e`)")
    for r in rows[:14]:
        print(f"  {r[0][:70]:70s} calls {r[1]:4d}  avg {r[3] / 1e3:9.1f} us")
# ---- kernel trace: the stats file averages ALL dispatches of a kernel (calibration launches, small input-generation
# launches); here the timed ones are separated: the last K dispatches of the headline kernel, and for every other kernel
# the dispatches of its LARGEST grid
trace = list((out / "stats").rglob("*kernel_trace.csv"))
if trace:
    disp = collections.defaultdict(list)
    with open(trace[0]) as fh:
        for r in csv.DictReader(fh):
            n = short(r["Kernel_Name"])
            if n.startswith(ours):
                disp[n.split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]),
                                               int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"])))
    prof = {}
    pj = out / "bench_profiled.json"
    if pj.exists() and pj.read_text().strip().startswith("{"):
        prof = json.loads(pj.read_text().strip().splitlines()[-1])
    lines = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-probe --no-e2e   (scripts/profile_round.sh {tag})",
             "# <tag>_kernel_stats.csv averages ALL dispatches of a kernel; this file separates the timed ones."]
    K = int(prof.get("steps", 20))
    for name, ds in sorted(disp.items(), key=lambda kv: -sum(d[1] for d in kv[1])):
        ds.sort()
        big = max(d[2] for d in ds)
        sel = [d for d in ds if d[2] == big]
        if name.startswith("vm_tile<6, true, true>"):
            # in start order: calibration (dxo_vm_output_alloc), the plain-allocation leg, warm-up, the K timed steps; the headline
            # block's launches are the last W + K of that grid group before the secondary legs start -> take the K before the
            # first dispatch of any secondary kernel
            first_secondary = min((dd[0][0] for nn, dd in disp.items() if nn.startswith(("mc_", "icnn_", "vm_field", "vm_tile<4", "heat_", "isihara", "tangent_", "adjoint", "assign"))), default=1 << 62)
            head = [d for d in ds if d[0] < first_secondary]
            B = int(prof.get("batches") or 1)
            timed = head[-B * K:]           # bench.py --no-side: nothing launches this kernel between the last timed batch and the secondary legs
            per_batch = [timed[b * K:(b + 1) * K] for b in range(B)]
            lines.append(f"{name}: {len(ds)} dispatches; the {B} x {K} timed steps (last {B * K} before the secondary legs): mean {sum(d[1] for d in timed) / len(timed) / 1e3:.1f} us "
                         f"(min {min(d[1] for d in timed) / 1e3:.1f}, max {max(d[1] for d in timed) / 1e3:.1f}); all others (calibration, warm-up): "
                         f"mean {sum(d[1] for d in head[:-B * K]) / max(len(head) - B * K, 1) / 1e3:.1f} us; VGPR {timed[0][3]}+{timed[0][4]} LDS {timed[0][5]} scratch {timed[0][6]}")
            lines.append("  per batch, mean us: " + " ".join(f"{sum(d[1] for d in pb) / max(len(pb), 1) / 1e3:.1f}" for pb in per_batch)
                         + "   (the bench line's kernel_ms_batches, HIP events of the same process: "
                         + " ".join(f"{x * 1e3:.1f}" for x in (prof.get("kernel_ms_batches") or [])) + ")")
            lines.append("  gaps between consecutive timed launches inside a batch, us (max per batch): "
                         + " ".join(f"{max((pb[k + 1][0] - pb[k][0] - pb[k][1]) / 1e3 for k in range(len(pb) - 1)):.1f}" for pb in per_batch if len(pb) > 1))
        else:
            lines.append(f"{name}: {len(ds)} dispatches, {len(sel)} at the largest grid ({big} threads): mean {sum(d[1] for d in sel) / len(sel) / 1e3:.1f} us "
                         f"(min {min(d[1] for d in sel) / 1e3:.1f}); VGPR {sel[0][3]}+{sel[0][4]} LDS {sel[0][5]} scratch {sel[0][6]}")
    if prof:
        lines.append(f"# the bench line printed by the SAME (profiled) process: roofline.kernel_ms_avg = {prof['roofline']['kernel_ms_avg'] * 1e3:.1f} us, "
                     f"frac {prof['roofline']['frac']:.4f}, ms_per_step {prof['ms_per_step'] * 1e3:.1f} us")
        sec = prof.get("secondary", {})
        for k, v in sec.items():
            if not isinstance(v, dict):
                continue
            if "ms_per_launch" in v:
                lines.append(f"#   secondary.{k}.ms_per_launch (HIP events, median) = {v['ms_per_launch'] * 1e3:.1f} us")
            for c, cv in (v.get("calls") or {}).items():
                lines.append(f"#   secondary.{k}.calls.{c}.ms_per_call (HIP events, median; fill + element kernel + node_sum) = {cv['ms_per_call'] * 1e3:.1f} us")
    (out / f"{tag}_kernel_trace_summary.txt").write_text("\n".join(lines) + "\n")
    print("== kernel trace summary\n" + "\n".join(lines))
bj = out / "bench.json"
if bj.exists() and bj.read_text().strip():
    (out / f"{tag}_bench_n1.json").write_text(bj.read_text())
    b = json.loads(bj.read_text())
    print(f"== bench: value {b['value']:.4g} qp/s, ms/step {b['ms_per_step']:.4f}, frac {b['roofline']['frac']:.4f}, kernel_ms_avg {b['roofline']['kernel_ms_avg']:.4f}")

# ---- headline HBM counters (the round-1 script does the arithmetic)
res = subprocess.run([sys.executable, str(pathlib.Path(__file__).with_name("summarize_pmc.py")), str(out), "--json", str(out / "traffic.json")],
                     capture_output=True, text=True)
(out / f"{tag}_pmc_summary.txt").write_text(res.stdout + res.stderr)
print("== headline HBM counters\n" + res.stdout)


def counters(dirs, match):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in (out / d).rglob("*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    key = match(short(r["Kernel_Name"]))
                    if key:
                        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


# ---- Mohr-Coulomb
mc = counters(["mc_valu", "mc_wave", "mc_fetch", "mc_write", "mcf_valu", "mcf_wave", "mcf_fetch", "mcf_write", "mcf_lane"],
              lambda n: "mc_newton" if "mc_newton" in n else "mc_classify" if "mc_classify" in n else "mc_fused" if "mc_fused" in n else None)
if mc:
    bench = None
    log = out / "mc_valu.log"
    for line in (log.read_text().splitlines() if log.exists() else []):
        if line.startswith("{"):
            bench = json.loads(line)
    derived = {}
    for k, c in mc.items():
        if "SQ_INSTS_VALU_FMA_F64" in c:
            derived[k + "_fp64_flop_per_launch"] = 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"])
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            derived[k + "_hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    bench_fused = None
    logf = out / "mcf_valu.log"
    for line in (logf.read_text().splitlines() if logf.exists() else []):
        if line.startswith("{"):
            bench_fused = json.loads(line)
    rec = {"counters": mc, "derived": derived, "bench_under_profiler": bench, "bench_under_profiler_single_kernel": bench_fused,
           "note": "flop = (ADD + MUL + 2 FMA wave-instructions) x 64 lanes, masked lanes counted; FETCH_SIZE doubled per MI355X_MICROARCH.md"}
    (out / f"{tag}_mc_pmc.json").write_text(json.dumps(rec, indent=1))
    if bench and "mc_newton_fp64_flop_per_launch" in derived:
        n, pl = bench["n"], bench["plastic_fraction"]
        flop = {"flop_per_plastic_point": derived["mc_newton_fp64_flop_per_launch"] / (n * pl),
                "flop_per_point_classify": derived.get("mc_classify_fp64_flop_per_launch", 0.0) / n,
                "measured": f"{tag}: rocprofv3 --pmc SQ_INSTS_VALU_{{ADD,MUL,FMA}}_F64 around scripts/bench_mc.py --variant 1 (classification and "
                            f"Newton as two kernels: the same per-point arithmetic as the default single kernel), {n} points, plastic fraction {pl:.4f}; "
                            "masked lanes counted (upper bound on useful flop)"}
        # the default single kernel priced with ITS OWN counters, and how many of the 64 lanes of its vector instructions did work
        cf = mc.get("mc_fused", {})
        if bench_fused and "mc_fused_fp64_flop_per_launch" in derived:
            fused = {"flop_per_launch": derived["mc_fused_fp64_flop_per_launch"], "points": bench_fused["n"],
                     "plastic_fraction": bench_fused["plastic_fraction"],
                     "measured": f"{tag}: SQ_INSTS_VALU_{{ADD,MUL,FMA}}_F64 of mc_fused itself (scripts/bench_mc.py --variant 2), masked lanes counted"}
            if cf.get("SQ_THREAD_CYCLES_VALU") and cf.get("SQ_ACTIVE_INST_VALU"):
                # SQ_ACTIVE_INST_VALU counts quad-cycles of a wave issuing vector instructions, SQ_THREAD_CYCLES_VALU the same cycles
                # weighted by the active lanes: their ratio / 64 (x 4 if the units differ — both forms are stored) is the mean
                # fraction of lanes that do work in the kernel's vector instructions
                raw = cf["SQ_THREAD_CYCLES_VALU"] / (64.0 * cf["SQ_ACTIVE_INST_VALU"])
                fused["lane_utilisation_raw_ratio"] = raw
                fused["lane_utilisation"] = raw if raw <= 1.0 else raw / 4.0
                fused["lane_utilisation_source"] = "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), pass mcf_lane (divided by 4 when the raw ratio exceeds 1: thread-cycles in cycles, wave-cycles in quad-cycles)"
            flop["fused"] = fused
        (out / "mc_flop.json").write_text(json.dumps(flop, indent=1))
        print("== Mohr-Coulomb:", json.dumps(derived), json.dumps(flop))

# ---- ICNN
ic = counters(["icnn_p1", "icnn_p2", "icnn_p3"], lambda n: "icnn_mfma" if "icnn_mfma" in n else None)
if ic:
    c = ic["icnn_mfma"]
    d = {}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        d["mfma_busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c.get("SQ_INSTS_MFMA", 1.0), 1.0)
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
        d["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    if "SQ_WAIT_INST_ANY" in c:
        d["wait_inst_any_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    (out / f"{tag}_icnn_pmc.json").write_text(json.dumps({"counters": ic, "derived": d}, indent=1))
    print("== ICNN:", json.dumps(d))

# ---- operand / fused field kernel
fd = counters(["field_fetch", "field_write", "field_sq1", "field_sq2", "field_tcc"],
              lambda n: "vm_field" if "vm_field" in n else "operand_eval" if "operand_eval" in n else "vm_tile" if "vm_tile" in n else None)
if fd:
    d = {}
    for k, c in fd.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d[k + "_hbm_MB_per_launch"] = (c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024) / 1e6
        if "TCC_MISS_sum" in c:
            d[k + "_tcc_miss_MB"] = c["TCC_MISS_sum"] * 128 / 1e6
    (out / f"{tag}_field_pmc.json").write_text(json.dumps({"counters": fd, "derived": d}, indent=1))
    print("== operand / vm_field:", json.dumps(d))

# ---- device-resident Newton iteration (tools/bench_device_loop.py): HBM bytes per call and SQ ratios per kernel
dl_files = {c: list((out / d).rglob("*counter_collection.csv")) for c, d in (("FETCH_SIZE", "dl_fetch"), ("WRITE_SIZE", "dl_write"))}
if all(dl_files.values()):
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    from tools import bench_secondary as bs

    f, w = bs.parse_counter_csv(dl_files["FETCH_SIZE"], "FETCH_SIZE"), bs.parse_counter_csv(dl_files["WRITE_SIZE"], "WRITE_SIZE")
    rec = {"hbm_bytes_per_call": {}, "note": "bytes = counter x 1024; FETCH_SIZE x2 for streaming kernels, x1 for the gather kernels (FETCH_X1 in "
                                             "tools/bench_secondary.py); node_sum / assign_store dispatches added to the call that launched them"}
    for (leg, path), (key, which) in bs.TRAFFIC_KEYS.items():
        if not leg.startswith(("device_loop", "assign")):
            continue
        fb, wb = bs._pick(f.get(key), which), bs._pick(w.get(key), which)
        if fb is not None and wb is not None:
            rec["hbm_bytes_per_call"][leg + ":" + "/".join(path[:-1] or ("call",))] = {"kernel": key, "fetch_corrected": fb, "write": wb, "total": fb + wb}
    names = ("tangent_apply<", "tangent_diag<", "tangent_cell<", "operand_adjoint_c8_mfma<", "operand_adjoint_c8<", "adjoint_cell_eps<", "node_sum<", "vm_field<", "vm_commit", "assign_owner", "assign_store", "assign_apply")
    sq = counters(["dl_sq1", "dl_sq2"], lambda n: next((k + n.split(k, 1)[1].split("(")[0] for k in names if k in n), None))
    ratios = {}
    for k, c in sq.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            ratios[k] = {r: round(c[r] / wc, 3) for r in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if r in c}
            if c.get("SQ_LDS_IDX_ACTIVE"):
                ratios[k]["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    rec["sq_counters_sampled"] = sq
    rec["sq_ratios_of_wave_cycles"] = ratios
    # fp64 arithmetic of the element kernels (their binding roof, not HBM): wave-instruction counts of the largest-grid dispatches,
    # flop = 64 x (ADD + MUL + 2 FMA) with masked lanes counted, over the kernel's duration in the kernel trace of the bench run
    fl = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in (out / "dl_flop").rglob("*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = short(r["Kernel_Name"])
                key = next((k + n.split(k, 1)[1].split("(")[0] for k in names if k in n), None)
                if key:
                    fl[key][r["Counter_Name"]].append((int(r.get("Grid_Size", 0)), float(r["Counter_Value"])))
    dur = {}
    if trace:
        for name, ds in disp.items():
            big = max(d[2] for d in ds)
            sel = [d[1] for d in ds if d[2] == big]
            dur[name] = sum(sel) / len(sel) / 1e6            # ms
    fp64 = {}
    for k, cs in fl.items():
        if "SQ_INSTS_VALU_FMA_F64" not in cs:
            continue
        def at_largest(c):
            vals = cs.get(c, [])
            if not vals:
                return 0.0
            big = max(g for g, _ in vals)
            sel = [v for g, v in vals if g == big]
            return sum(sel) / len(sel)
        flop = 64.0 * (at_largest("SQ_INSTS_VALU_ADD_F64") + at_largest("SQ_INSTS_VALU_MUL_F64") + 2.0 * at_largest("SQ_INSTS_VALU_FMA_F64"))
        if flop <= 0:
            continue
        e = {"fp64_flop_per_launch": flop, "trans_f64_wave_instructions": at_largest("SQ_INSTS_VALU_TRANS_F64")}
        mops = at_largest("SQ_INSTS_VALU_MFMA_MOPS_F64")          # the scatter of the hexahedral kernels on the matrix pipe (cell8_mfma.h): units of 512 flop,
        if mops > 0:                                               # padding rows / columns of the 16x16x4 tiles counted; NOT part of the vector-pipe figure above
            e["fp64_mfma_flop_per_launch"] = 512.0 * mops
        ms = next((v for n, v in dur.items() if n.startswith(k)), None)
        if ms:
            e.update({"kernel_ms_in_bench_trace": ms, "TFLOP_per_s": flop / ms / 1e9, "frac_of_78.6_TFLOP_per_s_fp64_vector_peak": flop / ms / 1e9 / 78.6})
        fp64[k] = e
    if fp64:
        # unversioned copy for tools/bench_device_loop.py (as mc_flop.json for the Mohr-Coulomb leg): the bench prices the consumer-side calls
        # with these counts when its meshes have the sizes they were counted on
        sizes = {}
        bjs = out / "bench.json"
        if bjs.exists() and bjs.read_text().strip():
            sec = json.loads(bjs.read_text()).get("secondary", {})
            sizes = {k: sec[k].get("points") for k in ("device_loop_q2hex", "device_loop_p2tri") if isinstance(sec.get(k), dict)}
        (out / "consumer_flop.json").write_text(json.dumps({"fp64_issued": fp64, "points": sizes, "measured": f"{tag}: rocprofv3 --pmc SQ_INSTS_VALU_{{ADD,MUL,FMA}}_F64 "
                                                           "around tools/bench_secondary.py --child (pass dl_flop of scripts/profile_round.sh); masked lanes counted"}, indent=1))
        rec["fp64_issued"] = fp64
        rec["fp64_note"] = ("flop = 64 x (ADD + MUL + 2 FMA) fp64 wave-instructions of the kernel's largest-grid dispatches (masked lanes counted: an upper bound on "
                            "useful work); duration = the kernel's mean in the kernel trace of the bench run (stats/), same sizes")
    (out / f"{tag}_device_loop_pmc.json").write_text(json.dumps(rec, indent=1))
    print("== device loop:", json.dumps(rec["hbm_bytes_per_call"]), json.dumps(ratios))
