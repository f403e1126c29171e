cd $GRAFT_REPO_ROOT
for l in 2 5; do
for a in 0 1 2 3 7 8; do echo "ABL=$a"; ./tools/conv_bench_abl$a $l 8 512 512 -1 -1 20; done
echo "1 block/CU:"; VSTAB_LDS_PAD=20000 ./tools/conv_bench_abl0 $l 8 512 512 -1 -1 20
VSTAB_LDS_PAD=20000 ./tools/conv_bench_abl3 $l 8 512 512 -1 -1 20
VSTAB_LDS_PAD=20000 ./tools/conv_bench_abl7 $l 8 512 512 -1 -1 20
done
