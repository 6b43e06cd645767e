export RW_BATCH=64 RW_ALGO=winograd4 RW_LAYERS=layer12,layer16
for v in 5 3; do export RW_WINO4_V=$v; echo "version $v product"; python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,95-125
for a in 2 4 6; do echo "v$v abl $a"; RW_HIP_LIB=$PWD/scripts/probe/lib_w4_abl$a.so timeout 100 python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,95-125; done; done
