"""Derived figures for the traversal and shading kernels from a pmc_passes.sh output directory."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["kernel"] for k in ("k_trace", "shade_bounce")):
        f = lambda k: float(r[k]) if r.get(k) else 0.0
        cyc = f("GRBM_GUI_ACTIVE") / 8
        wc = f("SQ_WAVE_CYCLES")
        print("==", r["kernel"], "calls", r["calls"], "total_us", r["total_us"])
        print("  active lanes per VALU instr      %.1f of 64" % (f("SQ_THREAD_CYCLES_VALU") / max(f("SQ_ACTIVE_INST_VALU"), 1)))
        print("  wave instrs: VALU %.3g SALU %.3g VMEM_RD %.3g LDS %.3g" % (f("SQ_INSTS_VALU"), f("SQ_INSTS_SALU"), f("SQ_INSTS_VMEM_RD"), f("SQ_INSTS_LDS")))
        print("  wave time: waiting %.2f, issue-stalled %.2f, issuing %.2f" % (f("SQ_WAIT_ANY") / wc, f("SQ_WAIT_INST_ANY") / wc, f("SQ_ACTIVE_INST_ANY") / wc))
        print("  resident waves per SIMD          %.2f" % (4 * wc / cyc / 1024))
        # a wave64 vector instruction occupies its SIMD for 2 cycles (measured on this kernel: 2.4 cycles per added
        # instruction, profiles/r02_bound_ablation.md); SIMD cycles = 4 SIMDs x 256 CUs x cycles
        print("  VALU issue, INSTS_VALU x 2.4 cycles over SIMD cycles %.2f   (x 2 cycles, the floor: %.2f)" % (
            f("SQ_INSTS_VALU") * 2.4 / 1024 / cyc, f("SQ_INSTS_VALU") * 2 / 1024 / cyc))
        if f("SQ_INST_CYCLES_VALU"):
            print("  SQ_INST_CYCLES_VALU x 4 over SIMD cycles %.2f, SQ_BUSY_CU_CYCLES share %.2f, TA address FIFO full %.3g, TA command FIFO full %.3g, LDS bank conflict share of LDS cycles %.2f" % (
                f("SQ_INST_CYCLES_VALU") * 4 / 1024 / cyc, f("SQ_BUSY_CU_CYCLES") / max(f("SQ_CYCLES"), 1), f("SQ_VMEM_TA_ADDR_FIFO_FULL"), f("SQ_VMEM_TA_CMD_FIFO_FULL"),
                f("SQ_LDS_BANK_CONFLICT") / max(f("SQ_LDS_IDX_ACTIVE"), 1)))
        print("  TA busy                          %.2f" % (f("TA_BUSY_avr") / cyc))
        print("  L1 accesses %.3g, L1->L2 %.3g (L1 hit %.2f), L2 hit %.2f, fabric reads %.3g, FETCH_SIZE %.1f GB, WRITE_SIZE %.2f GB" % (
            f("TCP_TOTAL_CACHE_ACCESSES_sum"), f("TCP_TCC_READ_REQ_sum"), 1 - f("TCP_TCC_READ_REQ_sum") / max(f("TCP_TOTAL_CACHE_ACCESSES_sum"), 1),
            f("TCC_HIT_sum") / max(f("TCC_REQ_sum"), 1), f("TCC_EA0_RDREQ_sum"), f("FETCH_SIZE") / 1e6, f("WRITE_SIZE") / 1e6))

# optional arguments <traffic.json> <workload> <source sha1>: write the fabric-side traffic per launch of the three
# hot kernels (FETCH_SIZE and WRITE_SIZE are KiB at the L2's fabric side, Infinity-Cache hits included; on gfx950
# FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled) for bench.py to report.  The hash of the kernel
# sources the counters were taken on is stored with them (pmc_passes.sh writes it to source_sha1.txt on the box):
# bench.py quotes an entry only while the tree still holds those sources.
if len(sys.argv) > 4:
    import json
    out, workload, sha = sys.argv[2], sys.argv[3], sys.argv[4]
    # the timed closest-hit instance (the counting instance runs once more, untimed), the any-hit instance of the
    # bounce levels, bounce shading
    wanted = {"k_trace<0>": "k_trace<0, false, false>", "k_trace<1>": "k_trace<1, true, false>", "k_shade_bounce": "k_shade_bounce"}
    try:
        table = json.load(open(out))
    except (OSError, ValueError):
        table = {}
    entry = {}
    for key, name in wanted.items():
        for r in rows:
            if name in r["kernel"] and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
                calls = float(r["calls"])
                fetch, write = float(r["FETCH_SIZE"]) * 1024, float(r["WRITE_SIZE"]) * 1024
                entry[key] = {"bytes_per_launch": (2 * fetch + write) / calls, "fetch_size_bytes_raw": fetch / calls,
                              "write_size_bytes": write / calls, "launches_profiled": int(calls), "source_sha1": sha,
                              "note": "2 x FETCH_SIZE + WRITE_SIZE per launch; fabric side of L2, Infinity Cache hits included"}
                print("traffic per launch of %s: %.2f GB" % (key, entry[key]["bytes_per_launch"] / 1e9))
                break
    table[workload] = entry
    json.dump(table, open(out, "w"), indent=1)
