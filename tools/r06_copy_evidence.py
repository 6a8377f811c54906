#!/usr/bin/env python3
"""Copy what tools/r06_final.sh left in gpurun_out/ into profiles/ (bench lines as the one JSON line; the stamped counter
profiles as profiles/*_latest.json) and print the figures the docs quote."""
import json, shutil, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out") + "/", os.path.join(R, "profiles") + "/"
def jsonline(src, dst):
    l = [x for x in open(G + src) if x.startswith("{")][-1]
    d = json.loads(l); open(P + dst, "w").write(l); return d
d = jsonline("r06_bench_b256_default.json", "r06_bench_b256_default.json")
d5 = jsonline("r06_bench_b256_5min.json", "r06_bench_b256_5min.json")
for f in ["r06_gpu_tests.log", "r06_parity_sweep_8448frames.txt", "r06_multipass_vs_reference_sandbox.txt", "r06_soak.txt", "r06_fine_timing.txt", "r06_osd_timing.txt", "r06_bp_timing.txt", "r06_kernel_stats_b256.txt",
          "r06_pmc_b256.txt", "r06_pmc_b4096_config2.txt", "r06_pmc_config3.txt", "r06_pmc_config4.txt", "r06_sq.txt", "r06_kernel_resources.txt"]:
    shutil.copy(G + f, P + f)
shutil.copy(G + "r06_pmc.json", P + "pmc_latest.json"); shutil.copy(G + "r06_sq.json", P + "sq_latest.json")
for c in (2, 3, 4):
    shutil.copy(G + f"r06_c{c}_pmc.json", P + f"pmc_config{c}_latest.json")
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "8d host audio", round(d["value_8d_host_audio"]), "over", d["value_8d_steps"], "steps; extra", round(d["extra_steps_frames_per_s"]))
print("other", {k: (round(v["value"]), round(v["true_decodes_per_frame"], 3), round(v["false_decodes_per_frame"], 4)) for k, v in d["other_configs"].items()})
r = d["roofline"]; v = d["roofline_valu"]
print("roofline", round(r["achieved"], 1), round(r["frac"], 4), "kernel_ms", round(r["kernel_ms"], 4), "traffic MB", round(r["traffic"] / 1e6, 1), r["traffic_stale"], "counter_frac", round(r["counter_frac"], 4), "whole", round(r["whole_path_frac"], 4))
print("valu", v["insts_per_launch"], round(v["frac"], 3), v["step_insts"], round(v["step_ms_sum_of_stages"], 3), round(v["step_frac"], 3), v["stale"])
print("cpu", round(d["cpu_baseline"]["value"], 1), round(d["cpu_baseline"]["all_cores"]["value"], 1))
print("stage_ms", d["stage_ms"]); print("gaps", d.get("step_gap_ms"))
print("5min", round(d5["value"]), d5["extra_steps"], round(d5["extra_steps_frames_per_s"]))
print("hash", json.load(open(P + "pmc_latest.json")).get("_source_hash"))
