#!/bin/bash
# One GPU call that regenerates the round's profiles (run on the GPU box from the repo root:  gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r05'):
# the default bench line, kernel trace + stats, phase timeline, decoder trace, the two PMC passes (HBM-side bytes per kernel) and the in-process A/B
# of the split-precision kernels.  Outputs land in gpurun_out/<tag>/; copy what is to be judged into profiles/ (names in profiles/README.md).
TAG=${1:-rNN}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err      # the driver's invocation
cp bench_detail.json $O/bench_detail.json          # the full record of THAT run (the profiling runs below overwrite bench_detail.json)
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o train -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-configs > $O/trace.log 2>&1
echo trace done
python tools/step_timeline.py $O/trace/train_kernel_trace.csv > $O/step_timeline.txt
python tools/decoder_trace.py $O/trace/train_kernel_trace.csv > $O/decoder_trace.txt
rm -f $O/trace/train_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs > $O/pmc_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs > $O/pmc_w.log 2>&1
echo write done
python tools/hbm_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv 3 > $O/hbm_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
python tools/xsplit_bench.py all > $O/xsplit_ab.txt 2>&1
echo ab done
timeout -k 10 120 python tools/xcd_sync_probe.py > $O/xcd_sync.txt 2>&1
echo sync probe done
