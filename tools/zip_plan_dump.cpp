// Prints the compile-time schedule of the zipped field backward (csrc/umhs_zip_plan.h):
//   g++ -std=c++17 -I unsupervised-hyperspectral-nerf_amd/csrc tools/zip_plan_dump.cpp -o /tmp/zip_plan_dump && /tmp/zip_plan_dump
#include <cstdio>

#include "umhs_zip_plan.h"

template <typename P, typename O>
static void dump(const char* name, const P& plan, const O& order, int nops, int nslots) {
  int m_in = 0, m_carry = 0, v_in = 0, v_carry = 0;
  for (int s = 0; s < 2 * nslots; ++s) {
    if (plan.m_at[s] >= 0) (s < nslots ? m_in : m_carry)++;
    if (plan.v_at[s] >= 0) (s < nslots ? v_in : v_carry)++;
  }
  std::printf("%s: ok=%d  %d ops, %d slots; MFMA ops in-tile %d carried %d; VALU ops in-tile %d carried %d; last slot %d\n", name, (int)plan.ok,
              nops, nslots, m_in, m_carry, v_in, v_carry, plan.last);
  static const char* jobs[] = {"TR", "PK", "dW_2", "dW_1", "dW_0", "dW_B1", "dW_B0", "j7", "j8", "j9", "j10", "j11"};
  for (int s = 0; s <= plan.last; ++s) {
    const int m = plan.m_at[s], v = plan.v_at[s];
    std::printf("%s%4d:", s == nslots ? "---- next tile ----\n" : "", s);
    if (v >= 0) std::printf("  V %s[%d]", jobs[order.op[v].job], order.op[v].idx);
    if (m >= 0) std::printf("  M %s[%d]", jobs[order.op[m].job], order.op[m].idx);
    std::printf("\n");
  }
}

int main() {
  dump("part 1", zp1::PLAN, zp1::ORDER, zp1::NOPS, zp1::NSLOTS);
  dump("part 0", zp0::PLAN, zp0::ORDER, zp0::NOPS, zp0::NSLOTS);
  return 0;
}
