"""Compose profiles/<tag>_pmc_neus.json from a tools/round_profile.sh run: the two pmc_summary.json files (traffic /
L2 passes and the SQ passes of the same mapper step) + the kernel durations of the same run's trace summary.
    python tools/compose_pmc_neus.py gpurun_out/r05d_final profiles/r05_pmc_neus.json
pmc_pass.sh keys its summary by the LAST 60 characters of a kernel name, so the kernels are picked by suffix."""
import json
import re
import sys

run, out = sys.argv[1], sys.argv[2]
tr = json.load(open(f"{run}/pmc_neus/pmc_summary.json"))
sq = json.load(open(f"{run}/pmc_neus_sq/pmc_summary.json"))
PICK = {"neus_encode_levels_kernel": "code_levels_kernel", "neus_point_kernel": "PKDF16_S4_PKDv4_ji",
        "neus_point_bwd_kernel<true,true>": None, "grid_bin_reduce_kernel": "grid_bin_reduce_kernel",
        "neus_mlp_bwd_kernel": "neus_mlp_bwd_kernel", "map_gram_kernel": "map_gram_kernel"}


def pick(d, name, frag):
    if frag is None:                    # the templated backward: "void neus_point_bwd_kernel<..>(..)" is cut at "void "
        return {c: v["median"] for c, v in d["void "].items()}
    ks = [k for k in d if frag in k]
    assert len(ks) == 1, (name, ks)
    return {c: v["median"] for c, v in d[ks[0]].items()}


dur = {}
for line in open(f"{run}/mapping_train_kernel_stats.md"):
    m = re.match(r"\| `([^`]+)` \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m:
        dur[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))


def us(frag):
    ks = [k for k in dur if frag in k]
    return dur[ks[0]][2] if ks else None


NP = 32768 * 72
res = {"command": "tools/round_profile.sh: PMC_GROUPS='FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum' and the SQ groups, "
                  "tools/pmc_pass.sh <out> _kernel -- python tools/profile_mapping.py train 3 (one counter group per rocprofv3 "
                  "--pmc pass, no trace domains; medians over the 3 launches; FETCH_SIZE / WRITE_SIZE in KB); composed by "
                  "tools/compose_pmc_neus.py from " + run,
       "workload": "fused mapper step, 32768 rays x 72 samples = 2,359,296 points, production mode: level-major forward "
                   "(neus_encode_levels_kernel + neus_point_kernel), binned fp16 table gradient"}
for name, frag in PICK.items():
    res[name] = pick(tr, name, frag)
x2 = {"neus_point_kernel", "grid_bin_reduce_kernel", "neus_mlp_bwd_kernel", "map_gram_kernel"}     # wide coalesced read streams
traffic = {n: ((2 if n in x2 else 1) * res[n]["FETCH_SIZE"] + res[n]["WRITE_SIZE"]) * 1000.0 for n in PICK}
res["traffic_bytes_per_launch@32768"] = traffic
res["traffic_convention"] = ("FETCH_SIZE x 2 for the kernels whose reads are wide coalesced streams (MI355X_MICROARCH.md, gfx950 "
                             "correction), x 1 for the encode kernel (4-byte gathers) and the backward's pass 1 (16-byte records + "
                             "scattered rows; as in r04)")
fwd = traffic["neus_encode_levels_kernel"] + traffic["neus_point_kernel"]
res["forward_total_bytes"] = fwd
res["traffic_over_algorithmic"] = {"forward (encode + point) / 524 B per point": fwd / (524.0 * NP),
                                   "neus_backward_points_binned / 1036 B per point":
                                       traffic["neus_point_bwd_kernel<true,true>"] / (1036.0 * NP)}
res["l2_hit_rate"] = {n: res[n]["TCC_HIT_sum"] / (res[n]["TCC_HIT_sum"] + res[n]["TCC_MISS_sum"])
                      for n in ("neus_encode_levels_kernel", "neus_point_kernel", "neus_point_bwd_kernel<true,true>")}
sqres = {}
for name, frag, dfrag in (("neus_point_bwd_kernel<true,true>", None, "neus_point_bwd_kernel"),
                          ("neus_encode_levels_kernel", "code_levels_kernel", "neus_encode_levels_kernel"),
                          ("neus_point_kernel", "PKDF16_S4_PKDv4_ji", "neus_point_kernelE"),
                          ("neus_mlp_bwd_kernel", "neus_mlp_bwd_kernel", "neus_mlp_bwd_kernel"),
                          ("grid_bin_reduce_kernel", "grid_bin_reduce_kernel", "grid_bin_reduce_kernel")):
    c = pick(sq, name, frag)
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                                   # summed over the 8 XCDs
    t = us(dfrag)
    waves = {"neus_encode_levels_kernel": None}.get(name, NP / 64.0)
    ent = {"counters": c, "kernel_cycles_per_xcd": cyc,
           "valu_busy_frac (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / kernel cycles)": c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc,
           "wave_wait_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]}
    if t:
        ent["kernel_us (trace of the same run)"] = t
        ent["effective_clock_GHz"] = cyc / t / 1e3
    if name in ("neus_point_bwd_kernel<true,true>", "neus_point_kernel"):
        ent["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / (NP / 64.0)
    sqres[name] = ent
res["sq"] = sqres
json.dump(res, open(out, "w"), indent=1)
b = sqres["neus_point_bwd_kernel<true,true>"]
print("forward traffic / algorithmic", res["traffic_over_algorithmic"], "L2", res["l2_hit_rate"])
print("bwd VALU per wave", b.get("valu_instructions_per_wave"), "busy", list(b.values())[2], "wait", list(b.values())[3])
