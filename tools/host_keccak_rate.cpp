#include "host_hash.hpp"
#include <stdio.h>
#include <chrono>
int main(){ printf("picked=%s\n", zk::host_keccak_impl()); uint64_t a[25]={1};
  for(int mode=1;mode<=4;mode++){ auto t0=std::chrono::steady_clock::now(); for(int i=0;i<2000000;i++) zk::host_keccak_permute(a,mode);
    double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count(); printf("variant %d: %.3f us/perm (%llu)\n",mode,dt/2e6*1e6,(unsigned long long)a[0]); } }
