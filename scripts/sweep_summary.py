#!/usr/bin/env python3
"""Summarise scripts/run_sweep.py outputs: mean evaluation episode_return at the 10 M-step evaluation (interval 61 of 122)
and at the end, mean +- std over seeds, plus the absolute metric.  usage: sweep_summary.py <out_dir> [> table.md]"""
import glob
import json
import os
import sys

import numpy as np

out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "json", "*", "metrics.json"))):
    tag = os.path.basename(os.path.dirname(f))
    scen, seed = tag.rsplit("_s", 1)
    if len(sys.argv) > 2 and scen not in sys.argv[2:]:
        continue
    d = json.load(open(f))
    run = next(iter(next(iter(next(iter(next(iter(d.values())).values())).values())).values()))
    steps = sorted((int(k.split("_")[1]), v) for k, v in run.items() if k.startswith("step_"))
    curve = [(v["step_count"], v["mean_episode_return"][0]) for _, v in steps if "mean_episode_return" in v]
    rows.setdefault(scen, []).append((int(seed), curve, run.get("absolute_metrics", {}).get("mean_episode_return", [float("nan")])[0],
                                      np.mean([v["steps_per_second"][0] for _, v in steps if "steps_per_second" in v])))
print("| scenario | seeds | eval return at half the run (mean +- std over seeds) | eval return at the end of the run | absolute metric (best params, 320 episodes) | first evaluation | evaluator env-steps/s |")
print("|---|---|---|---|---|---|---|")
for scen, runs in rows.items():
    def at(frac):
        vals = []
        for _, curve, _, _ in runs:
            tgt = curve[-1][0] * frac
            vals.append(min(curve, key=lambda c: abs(c[0] - tgt))[1])
        return np.array(vals)
    mid, end, first = at(0.5), at(1.0), np.array([c[1][0][1] for c in runs])
    ab = np.array([r[2] for r in runs])
    T_end = runs[0][1][-1][0] / 1e6
    print(f"| {scen} | {len(runs)} | {mid.mean():.2f} +- {mid.std():.2f} (at {T_end / 2:.1f} M steps) | {end.mean():.2f} +- {end.std():.2f} (at {T_end:.1f} M steps) | "
          f"{np.nanmean(ab):.2f} +- {np.nanstd(ab):.2f} | {first.mean():.2f} | {np.mean([r[3] for r in runs]):.0f} |")
