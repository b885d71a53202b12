# A/B of the paired-waves form of prompt attention (library built with -DSW_PREFILL_EXPERIMENT)
for p in 0 1; do echo "## SW_PREFILL_PAIRS=$p"; SW_PREFILL_PAIRS=$p timeout -k 10 200 python -u tools/bench_prefill_attention.py 2>&1 | grep -v amdgpu.ids || exit 1; done
