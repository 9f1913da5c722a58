// C entry of the substep planner for the CPU test (lib/libudcplan.so, built by g++; not part of libudcore's interface).
#include "udc_plan.h"
extern "C" int udc_plan_fields(void) { return (int)(sizeof(PlanIn) / sizeof(int)); }
extern "C" int udc_plan_outputs(void) { return (int)(sizeof(Plan) / sizeof(int)); }
extern "C" void udc_plan_substep(const int *in, int *out) {
  PlanIn a;
  int *pa = reinterpret_cast<int *>(&a);
  for (int q = 0; q < (int)(sizeof(PlanIn) / sizeof(int)); ++q) pa[q] = in[q];
  const Plan p = plan_substep(a);
  const int *pp = reinterpret_cast<const int *>(&p);
  for (int q = 0; q < (int)(sizeof(Plan) / sizeof(int)); ++q) out[q] = pp[q];
}
// n rows of PlanIn -> n rows of Plan
extern "C" void udc_plan_batch(const int *in, long n, int *out) {
  const int ni = (int)(sizeof(PlanIn) / sizeof(int)), no = (int)(sizeof(Plan) / sizeof(int));
  for (long r = 0; r < n; ++r) udc_plan_substep(in + r * ni, out + r * no);
}
