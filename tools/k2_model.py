"""Issue-slot model of K2 (stft_reassigned_4096_tri_kernel<2>) from its ISA and the per-instruction cycle tables of
/opt/skills/guides/MI355X_MICROARCH.md ("commit an issue-slot model that accounts for >= 90 % of the kernel").

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -x hip --cuda-device-only -S \
          openmeters_amd/csrc/stft4096_tri_kernels.hip -o /tmp/tri.s
    python tools/k2_model.py /tmp/tri.s [kernel_ms] [profiles/rNN_bench_summary.txt]

The kernel is straight-line per workgroup (uniform branches pick the ring-addressing path and the odd tail): the static census of the
main path equals the dynamic count per wavefront (checked against SQ_INSTS_VALU / SQ_INSTS_LDS of the profile when given)."""
import collections
import re
import sys

CLOCK_GHZ = 2.4
CUS, SIMDS = 256, 1024
FRAMES = 65536                       # columns per launch of the benchmark workload
WGS = FRAMES // 2                    # one workgroup per pair of columns
WAVES = WGS * 4

# LDS-pipe cycles per wave-instruction (guide, "LDS [CDNA4]": LDS-array cycles for loads, VGPR -> LDS transfer for stores)
LDS_CYCLES = {"ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read2_b32": 4, "ds_read2_b64": 8, "ds_read2st64_b64": 8,
              "ds_write_b32": 4, "ds_write_b64": 6, "ds_write2_b32": 6, "ds_write2_b64": 13, "ds_write2st64_b64": 13, "ds_write_b128": 13,
              "ds_bpermute_b32": 4, "ds_swizzle_b32": 4}
# VALU issue cycles per wave64 instruction on a SIMD (4 passes of 16 lanes; transcendentals and f64 are quarter / half rate)
SLOW_VALU = {"v_rcp_f32": 16, "v_log_f32": 16, "v_exp_f32": 16, "v_sqrt_f32": 16, "v_rsq_f32": 16, "v_sin_f32": 16, "v_cos_f32": 16}


def census(path, kernel="_ZN3omx31stft_reassigned_4096_tri_kernelILi2EEEvNS_12StftFastArgsE"):
    text = open(path).read()
    start = text.index(kernel + ":")
    end = text.index("s_endpgm", start)
    # the wrapped-ring path (`direct` false: a window that straddles the ring's end) and the silent exit are cold: drop blocks that
    # hold global_load_dword x2 pairs of the per-element path by counting the buffer_load form only when both exist
    cnt = collections.Counter()
    for ln in text[start:end].split("\n"):
        ln = ln.strip()
        if not ln or ln[0] in ";." or ln.endswith(":"):
            continue
        cnt[ln.split()[0]] += 1
    return cnt


def main():
    cnt = census(sys.argv[1])
    kernel_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.38
    valu = {k: v for k, v in cnt.items() if k.startswith("v_")}
    lds = {k: v for k, v in cnt.items() if k.startswith("ds_")}
    salu = sum(v for k, v in cnt.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and not k.startswith("s_barrier") and k != "s_nop")
    vmem = {k: v for k, v in cnt.items() if k.startswith("buffer_") or k.startswith("global_")}
    barriers = cnt.get("s_barrier", 0)
    n_valu = sum(valu.values())
    valu_cycles = sum(v * SLOW_VALU.get(k.replace("_e32", "").replace("_e64", ""), 4) for k, v in valu.items())
    lds_cycles = sum(v * LDS_CYCLES.get(k, 4) for k, v in lds.items())
    print(f"static census of one wavefront's path: VALU {n_valu}, LDS {sum(lds.values())}, SALU {salu}, VMEM {sum(vmem.values())}, s_barrier {barriers}")
    print("LDS instructions:", ", ".join(f"{k} x{v}" for k, v in sorted(lds.items(), key=lambda kv: -kv[1])))
    print("VMEM instructions:", ", ".join(f"{k} x{v}" for k, v in sorted(vmem.items(), key=lambda kv: -kv[1])))
    waves_per_simd = WAVES / SIMDS
    waves_per_cu = WAVES / CUS
    t_valu = valu_cycles * waves_per_simd / (CLOCK_GHZ * 1e6)          # ms: every SIMD issues its waves' VALU instructions one at a time
    t_lds = lds_cycles * waves_per_cu / (CLOCK_GHZ * 1e6)              # ms: ONE LDS pipe per CU serves all twelve resident wavefronts
    print(f"VALU issue: {valu_cycles} cycles per wavefront x {waves_per_simd:.1f} wavefronts per SIMD = {t_valu:.3f} ms at {CLOCK_GHZ} GHz")
    print(f"LDS pipe:   {lds_cycles} cycles per wavefront x {waves_per_cu:.1f} wavefronts per CU   = {t_lds:.3f} ms")
    print(f"sum {t_valu + t_lds:.3f} ms, max {max(t_valu, t_lds):.3f} ms, measured kernel {kernel_ms:.3f} ms -> "
          f"VALU busy {t_valu / kernel_ms * 100:.0f} %, LDS pipe busy {t_lds / kernel_ms * 100:.0f} %, "
          f"sum / measured = {(t_valu + t_lds) / kernel_ms * 100:.0f} %")
    if len(sys.argv) > 3:
        prof = open(sys.argv[3]).read()
        def grab(name):
            m = re.search(r"stft_reassigned_4096_tri_kernel<2>\s+" + name + r"\s+mean=([0-9.e+]+)", prof)
            return float(m.group(1)) if m else None
        waves = grab("SQ_WAVES") or WAVES
        for name, mine in (("SQ_INSTS_VALU", n_valu), ("SQ_INSTS_LDS", sum(lds.values())), ("SQ_INSTS_SALU", None)):
            v = grab(name)
            if v:
                print(f"profile {name}: {v:.4g} per launch = {v / waves:.0f} per wavefront" + (f" (static census {mine}: the census includes the cold wrapped-ring path)" if mine else ""))
        dyn_valu = grab("SQ_INSTS_VALU")
        if dyn_valu:
            t_dyn = dyn_valu / SIMDS * 4 / (CLOCK_GHZ * 1e6)
            print(f"VALU issue from the DYNAMIC count: {dyn_valu / waves:.0f} instructions x 4 cycles x {waves_per_simd:.1f} wavefronts per SIMD = {t_dyn:.3f} ms at {CLOCK_GHZ} GHz "
                  f"-> VALU busy {t_dyn / kernel_ms * 100:.0f} %, VALU + LDS pipe = {(t_dyn + t_lds):.3f} ms = {(t_dyn + t_lds) / kernel_ms * 100:.0f} % of the measured kernel")
        wc = grab("SQ_WAVE_CYCLES")
        if wc:
            print("wave cycles (SQ_WAVE_CYCLES = ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY, guide):")
            for name, label in (("SQ_ACTIVE_INST_ANY", "issuing an instruction"), ("SQ_ACTIVE_INST_VALU", "  of which VALU"), ("SQ_ACTIVE_INST_LDS", "  of which LDS"),
                                ("SQ_WAIT_INST_ANY", "stalled at issue"), ("SQ_WAIT_INST_LDS", "  of which at the LDS queue"), ("SQ_WAIT_ANY", "parked at s_waitcnt / s_barrier")):
                v = grab(name)
                if v:
                    print(f"  {label:34s} {v / wc * 100:5.1f} %")
            conf, idx = grab("SQ_LDS_BANK_CONFLICT"), grab("SQ_LDS_IDX_ACTIVE")
            if conf and idx:
                print(f"  LDS bank-conflict cycles / LDS-array cycles {conf / idx * 100:.1f} %")


if __name__ == "__main__":
    main()
