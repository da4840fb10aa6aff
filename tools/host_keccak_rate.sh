cd $GRAFT_REPO_ROOT
C=/opt/rocm/lib/llvm/bin/clang++
$C -O3 -std=c++17 -Izigz_amd/csrc tools/host_keccak_rate.cpp zigz_amd/csrc/host_hash.cpp zigz_amd/csrc/host_keccak_avx512.cpp zigz_amd/csrc/host_keccak_bmi.cpp zigz_amd/csrc/host_keccak_avx512vl.cpp -o /tmp/hkr && /tmp/hkr
g++ -O3 -std=c++17 -Izigz_amd/csrc tools/host_keccak_rate.cpp zigz_amd/csrc/host_hash.cpp zigz_amd/csrc/host_keccak_avx512.cpp zigz_amd/csrc/host_keccak_bmi.cpp zigz_amd/csrc/host_keccak_avx512vl.cpp -o /tmp/hkr2 && /tmp/hkr2
$C -O3 -march=native -std=c++17 -Izigz_amd/csrc tools/host_keccak_rate.cpp zigz_amd/csrc/host_hash.cpp zigz_amd/csrc/host_keccak_avx512.cpp zigz_amd/csrc/host_keccak_bmi.cpp zigz_amd/csrc/host_keccak_avx512vl.cpp -o /tmp/hkr3 && /tmp/hkr3
grep -m1 "model name" /proc/cpuinfo; grep -m1 "cpu MHz" /proc/cpuinfo; grep -o "avx512[a-z_0-9]*" /proc/cpuinfo | sort -u | tr '\n' ' '
