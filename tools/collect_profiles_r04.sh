#!/bin/bash
# Copies the summaries of tools/profile_r04.sh + tools/profile_extra.sh r04 (+ the plain bench repeated behind them, gpurun_out/bench_final.log,
# whose line carries roofline.traffic) from gpurun_out/ into profiles/r04/ and adds the C5 traffic key to traffic.json.
set -eu
newest() { ls -t $1 | head -1; }
G=gpurun_out; D=profiles/r04; X=$G/prof_r04x
mkdir -p $D
cp $G/prof_r04/traffic.json $G/prof_r04/bench_under_rocprof.log $G/prof_r04/pmc_instruction_mix.log $G/prof_r04/bench_kernel_stats.csv $D/
cp $G/bench_final.log $D/bench_plain.log
cp $(newest "$G/prof_r04/pmc_fetch/runc/*_counter_collection.csv") $D/pmc_fetch.csv
cp $(newest "$G/prof_r04/pmc_write/runc/*_counter_collection.csv") $D/pmc_write.csv
cp $X/bench_c5_b16.log $X/bench_c5_b16_rocprof.log $X/wave_mfma_probe.log $X/agpr_tile_probe.log $X/register_bench.log $D/
cp $(newest "$X/c5_stats/runc/*_kernel_stats.csv") $D/c5_b16_kernel_stats.csv
cp $(newest "$X/c5_fetch/runc/*_counter_collection.csv") $D/c5_pmc_fetch.csv
cp $(newest "$X/c5_write/runc/*_counter_collection.csv") $D/c5_pmc_write.csv
cp $X/nrsfm_plain.log $D/nrsfm_bench.log
cp $(newest "$X/nrsfm_stats/runc/*_kernel_stats.csv") $D/nrsfm_kernel_stats.csv
python tools/scratch_report.py > $D/scratch_report.txt 2>&1 || true
python - <<'PY'
import csv, json
def tot(f, counter, kernel):
    s = 0.0; n = 0
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"]); n += 1
    return s, n
f, nf = tot('profiles/r04/c5_pmc_fetch.csv', 'FETCH_SIZE', 'sft_spec_kernel')
w, nw = tot('profiles/r04/c5_pmc_write.csv', 'WRITE_SIZE', 'sft_spec_kernel')
runs = 3   # the C5 PMC passes run 2 steps + 1 warm-up
tj = json.load(open('profiles/r04/traffic.json'))
tj["C5_B16"] = {"fetch_kib": f / runs, "write_kib": w / runs, "bytes_per_launch": int((2 * f + w) * 1024 / runs), "launches_in_pass": [nf, nw],
                "what": "all sft_spec_kernel<8> launches of ONE step of 16 C5 problems (latency mode: one launch per round)"}
json.dump(tj, open('profiles/r04/traffic.json', 'w'), indent=1)
print("C5_B16 GB per step", tj["C5_B16"]["bytes_per_launch"] / 1e9)
PY
ls $D
