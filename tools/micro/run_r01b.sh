./tools/micro/lds_align
bash tools/bench_variants.sh
