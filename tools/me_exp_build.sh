# builds of me_kernels.hip with extra stage marks (-DME_EXP=10: the finer stamps tools/me_phase_profile.py reads with ME_EXP_STAMPS=1): tools/_exp/lib<N>.so, used through SVT_PRODUCT_LIB
set -e
cd "$(dirname "$0")/../svt-hevc_amd"
mkdir -p ../tools/_exp
for n in "$@"; do
  /opt/rocm/bin/hipcc -DME_EXP=$n --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -c csrc/me_kernels.hip -o ../tools/_exp/me_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/_exp/lib$n.so ../tools/_exp/me_$n.o $(ls build/*.o | grep -v me_kernels.o)
done
ls -la ../tools/_exp/*.so
