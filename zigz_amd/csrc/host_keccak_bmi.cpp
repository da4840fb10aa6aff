// Portable-formulation Keccak-f[1600] compiled for BMI1/BMI2 (andn for chi, rorx for rho): same round
// macro as the generic build, different instruction selection.  Chosen at load time by calibration
// (host_hash.cpp) when the CPU supports it and it is the fastest variant.
#include <stdint.h>

#include "keccak.hpp"

namespace zk {

__attribute__((target("bmi,bmi2"))) void keccak_f1600_bmi(uint64_t st[25]) {
    for (int r = 0; r < 24; r++) ZK_KECCAK_ROUND(st, KECCAK_RC[r]);
}

bool cpu_has_bmi2() { return __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2"); }

}  // namespace zk
