// Does gfx950 keep PROGRAM ORDER between a DS instruction and a FLAT instruction of the same wave that touch the same LDS address?
// (FLAT goes through the vector-memory address path before it reaches the LDS; DS goes to the LDS queue directly.  The compiler
//  inserts s_waitcnt for REGISTER dependencies only.)  Six orderings, each N times per lane, mismatches counted:
//   0 RAW  ds_write  then flat_load    (load must see the new value)
//   1 RAW  flat_store then ds_read     (read must see the new value)
//   2 WAR  flat_load then ds_write     (load must see the OLD value)
//   3 WAR  ds_read   then flat_store   (read must see the OLD value)
//   4 WAW  ds_write(A) then flat_store(B) -> B must stay
//   5 WAW  flat_store(A) then ds_write(B) -> B must stay
// flat_ds_order [iterations] [waves per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__global__ void k_order(int n, unsigned* bad)
{
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const int tid = threadIdx.x;
  u64* slot = lds + tid;  // one 8-byte slot per lane
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) u64*)slot;
  u64* fp = slot;  // generic (flat) pointer to the same slot
  unsigned cnt[6] = { 0, 0, 0, 0, 0, 0 };
  for (int it = 0; it < n; ++it)
  {
    const u64 oldv = 0x1111000000000000ULL + (u64)it * 7 + tid, newv = 0x2222000000000000ULL + (u64)it * 13 + tid, thirdv = newv ^ 0xFFFFULL;
    u64 got;
    // 0: ds_write then flat_load
    *slot = oldv;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("ds_write_b64 %1, %2\n flat_load_dwordx2 %0, %3\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "=&v"(got) : "v"(laddr), "v"(newv), "v"(fp) : "memory");
    cnt[0] += got != newv;
    // 1: flat_store then ds_read
    *slot = oldv;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("flat_store_dwordx2 %3, %2\n ds_read_b64 %0, %1\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "=&v"(got) : "v"(laddr), "v"(newv), "v"(fp) : "memory");
    cnt[1] += got != newv;
    // 2: flat_load then ds_write
    *slot = oldv;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("flat_load_dwordx2 %0, %3\n ds_write_b64 %1, %2\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "=&v"(got) : "v"(laddr), "v"(newv), "v"(fp) : "memory");
    cnt[2] += got != oldv;
    // 3: ds_read then flat_store
    *slot = oldv;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("ds_read_b64 %0, %1\n flat_store_dwordx2 %3, %2\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "=&v"(got) : "v"(laddr), "v"(newv), "v"(fp) : "memory");
    cnt[3] += got != oldv;
    // 4: ds_write(new) then flat_store(third)
    *slot = oldv;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("ds_write_b64 %0, %1\n flat_store_dwordx2 %3, %2\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(laddr), "v"(newv), "v"(thirdv), "v"(fp) : "memory");
    got = *(volatile u64*)slot;
    cnt[4] += got != thirdv;
    // 5: flat_store(new) then ds_write(third)
    *slot = oldv;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("flat_store_dwordx2 %3, %1\n ds_write_b64 %0, %2\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(laddr), "v"(newv), "v"(thirdv), "v"(fp) : "memory");
    got = *(volatile u64*)slot;
    cnt[5] += got != thirdv;
  }
  for (int k = 0; k < 6; ++k)
    if (cnt[k])
      atomicAdd(&bad[k], cnt[k]);
}
int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 2000;
  const int waves = argc > 2 ? atoi(argv[2]) : 4;
  unsigned* bad;
  (void)hipMalloc(&bad, 6 * sizeof(unsigned));
  (void)hipMemset(bad, 0, 6 * sizeof(unsigned));
  hipLaunchKernelGGL(k_order, dim3(256), dim3(64 * waves), 64 * waves * 8, 0, n, bad);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[6];
  (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = { "RAW ds_write->flat_load", "RAW flat_store->ds_read", "WAR flat_load->ds_write", "WAR ds_read->flat_store", "WAW ds_write->flat_store", "WAW flat_store->ds_write" };
  printf("%s; %d iterations x %d lanes x 256 workgroups\n", hipGetErrorString(e), n, 64 * waves);
  for (int k = 0; k < 6; ++k)
    printf("  %-28s mismatches %u\n", names[k], h[k]);
  return 0;
}
