// Scratch probe: host fq_mul / fq_invert cost per compiler (g++ vs ROCm clang++): g++ -O2 -Iinclude -Ispartan_amd/csrc bench/host_arith_probe.cc
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstring>
#include "field.hpp"
using namespace sp;
int main(){ Fq a = fq_from_u64(123456789), b = fq_from_u64(987654321); auto t0=std::chrono::steady_clock::now(); for(int i=0;i<10000000;i++) a = fq_mul(a,b); double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count(); printf("fq_mul %.1f ns %llx\n", dt/1e7*1e9,(unsigned long long)a.l[0]);
 t0=std::chrono::steady_clock::now(); for(int i=0;i<20000;i++) a = fq_invert(a); dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count(); printf("fq_invert %.1f ns %llx\n", dt/2e4*1e9,(unsigned long long)a.l[0]); }
