cd $GRAFT_REPO_ROOT
./tools/conv_bench_abl0 13 8 512 512 1 1 20
./tools/conv_bench_abl0 13 8 512 512 3 1 20
./tools/conv_bench_abl0 13 8 512 512 3 2 20
./tools/conv_bench_abl0 13 8 512 512 1 1 20
./tools/conv_bench_abl0 13 8 512 512 3 1 20
