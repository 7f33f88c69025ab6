for v in base "" pl8 base "" pl8; do
  if [ -z "$v" ]; then unset PPASR_HIP_LIB; name=pl4; else export PPASR_HIP_LIB=tools/_ts/lib_$v.so; name=$v; fi
  python bench.py --gemm f16x3 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['classes']; print('$name', d['ms_per_step'], c['conv2/f16x3']['ms_per_step'], c['k_conv1']['ms_per_step'], c['k_conv_ffn<15>+next/f16x3']['ms_per_step'])"
done
