for shape in "13824 5120" "7680 5120" "5120 2560" "5120 6912"; do
  for M in 160 256; do
    echo "== M=$M N,K=$shape"
    timeout -k 10 60 tools/bin/quant_exp $M $shape "kq" | grep -v "t256\|t64\|ks3\|ks4" || exit 1
  done
done
