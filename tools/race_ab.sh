for lib in lightglue_amd/liblightglue_amd.so lightglue_amd/liblightglue_amd_noslp.so lightglue_amd/_ab/liblightglue_amd_r1.so; do
for i in 1 2 3; do echo $lib $i; LIGHTGLUE_AMD_LIB=$PWD/$lib python tools/diag_proj_race2.py 2>&1 | grep -v amdgpu.ids | grep "run 0"; done; done
