# the eight-rank bench over the RCCL double, N times (a flake hunt: round 6 found a lost-tile race of k_fuse<.., HEAD> with it)
cd $GRAFT_REPO_ROOT
N=${1:-6}
FAKE=$(python -c "import sys; sys.path.insert(0,'tests'); import test_parallel as t; print(t._build_fake_rccl())")
for i in $(seq 1 $N); do
timeout 900 python bench.py --gpus 8 --steps 10 --warmup 3 --repeats 2 --c4-frames 16 --rccl-double $FAKE --raycast-reps 0 --no-staged --rank-timeout 800 > gpurun_out/bench8_$i.out 2> gpurun_out/bench8_$i.err; echo "run $i rc $?"
done
