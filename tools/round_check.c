/* Exhaustive check (all 2^32 floats, ~20 s): trunc(x + copysign(0.5 - 2^-25, x)) == roundf(x) bit for bit -- the form gsdf_roundf
 * (csrc/gsdf_math.h) uses on the GPU.  gcc -O2 -fno-fast-math -ffp-contract=off tools/round_check.c -lm && ./a.out  ->  bad=0 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float alt(float x){ float h=copysignf(0.49999997f,x); return truncf(x+h); }
int main(){ uint64_t bad=0; for(uint64_t i=0;i<(1ull<<32);++i){ uint32_t u=(uint32_t)i; float x; memcpy(&x,&u,4); float a=roundf(x), b=alt(x); uint32_t ua,ub; memcpy(&ua,&a,4); memcpy(&ub,&b,4); if(ua!=ub && !(isnan(a)&&isnan(b))){ if(bad<10) printf("x=%a round=%a alt=%a\n",x,a,b); ++bad; } } printf("bad=%llu\n",(unsigned long long)bad); return 0; }
