for M in 128 256; do for K in 1280 2560 5120 10240 20480; do echo "== M=$M K=$K"; timeout -k 10 60 tools/bin/quant_exp $M 13824 $K "kq4 ks1" || exit 1; done; done
