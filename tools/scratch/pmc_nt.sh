export TMPDIR=/tmp
for v in "" _ntl _ntls; do
  for c in FETCH_SIZE WRITE_SIZE; do
    LIGHTGLUE_AMD_LIB=$PWD/lightglue_amd/liblightglue_amd$v.so rocprofv3 --kernel-trace --pmc $c -d gpurun_out/nt/$c$v -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    echo "== lib$v $c"; python tools/rocpd_pmc.py $(find gpurun_out/nt/$c$v -name "*.db" | head -1) | grep "tail_kernelILi3ELi[12]"
  done
done
find gpurun_out/nt -name "*.db" -delete
