cd $GRAFT_REPO_ROOT
for l in 2 1 5; do for st in 0 16 32 64 0 32; do echo -n "st=$st "; ./tools/conv_bench_st$st $l 8 512 512 -1 -1 20; done; done
