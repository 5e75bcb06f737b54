mkdir -p gpurun_out
R=$PWD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -x --timeout=600 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest.log
python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
for w in roi_align_rotated_bwd box_iou_rotated; do python bench.py --workload $w --steps 50 --warmup 5 > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log; done
python bench.py --workload nms_rotated --rois 8576 --steps 20 --warmup 3 > gpurun_out/bench_nms.log 2>&1; tail -1 gpurun_out/bench_nms.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fwd -o fwd -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_fwd.log 2>&1; echo "prof rc=$?"
ls $R/gpurun_out/prof_fwd | head
