for shape in "13824 5120" "7680 5120" "5120 2560" "5120 6912"; do
  for M in 8 16 32; do
    echo "== M=$M N,K=$shape"
    timeout -k 10 60 tools/bin/quant_exp $M $shape "t32" || exit 1
  done
done
