cd $GRAFT_REPO_ROOT
for ks in 1 2 3 4 6; do ./tools/conv_bench_abl0 13 8 512 512 1 $ks 20; done
for ks in 1 2 3; do ./tools/conv_bench_abl0 13 8 512 512 0 $ks 20; done
for ks in 1 2 3; do ./tools/conv_bench_abl0 12 8 512 512 0 $ks 20; done
for ks in 2 3 4 6; do ./tools/conv_bench_abl0 11 8 512 512 0 $ks 20; done
for ks in 1 2 3; do ./tools/conv_bench_abl0 4 8 512 512 0 $ks 20; done
for ks in 4 6 8 12; do ./tools/conv_bench_abl0 6 8 512 512 0 $ks 20; done
for ks in 8 12 16 24; do ./tools/conv_bench_abl0 8 8 512 512 0 $ks 20; done
