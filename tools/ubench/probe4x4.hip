// probes the lane layout and the CBSZ/ABID broadcast semantics of v_mfma_f64_4x4x4_4b_f64 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID>
__global__ void k_probe(int* out)  // out[la*64+lb] = lane of the (single) nonzero D, or -1 ; -2 if several
{
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
    {
      const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (l == 0)
      {
        int r = -1;
        if (m)
        {
          r = __builtin_ctzll(m);
          if (m & (m - 1)) r = -2 - __builtin_popcountll(m);
        }
        out[la * 64 + lb] = r;
      }
    }
}
template <int CBSZ, int ABID>
void run(const char* name)
{
  int* d; hipMalloc(&d, 64 * 64 * sizeof(int));
  hipLaunchKernelGGL((k_probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  static int h[64 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("== %s\n", name);
  for (int la = 0; la < 64; ++la)
  {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] != -1) printf(" (B%d->D%d)", lb, h[la * 64 + lb]);
    printf("\n");
  }
  hipFree(d);
}
int main()
{
  run<0, 0>("cbsz=0 abid=0");
  run<1, 0>("cbsz=1 abid=0");
  run<1, 1>("cbsz=1 abid=1");
  run<2, 1>("cbsz=2 abid=1");
  return 0;
}
