# same-box A/B: the fused attention's score contraction in fp32 (lib_nosplit: the commit before) vs on the fp16 x3 split
for v in nosplit "" nosplit ""; do
  if [ -z "$v" ]; then unset PPASR_HIP_LIB; name=split; else export PPASR_HIP_LIB=tools/_ts/lib_$v.so; name=$v; fi
  python bench.py --gemm f16x3 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['classes']; print('$name', d['ms_per_step'], 'attn', c['k_attn_out_glu/f16x3']['ms_per_step'], 'layer', c['k_conv_ffn<15>+next/f16x3']['ms_per_step'])"
done
