# 8-k-part tile variants of the int4 GEMM against the shipped choice (tools/bin/quant_exp, built with -DSW_QUANT_EXPERIMENT)
for shape in "13824 5120" "7680 5120" "5120 2560" "5120 6912"; do
  for M in 48 64 96 128; do
    echo "== M=$M N,K=$shape"
    timeout -k 10 60 tools/bin/quant_exp $M $shape "kq" | grep -v "t128 kq2\|t256" || exit 1
  done
done
