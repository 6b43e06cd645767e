export RW_BATCH=64 RW_ALGO=winograd4 RW_LAYERS=layer12,layer16,layer18 RW_WINO4_V=2
echo product; python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,95-125
for a in 1 2 4 8 16 6 7 23 31; do echo "abl $a"; RW_HIP_LIB=$PWD/scripts/probe/lib_w4_abl$a.so timeout 100 python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,95-125; done
