// temporary stubs (replaced as the rows of SURVEY §8 are filled in)
#include "kg_internal.h"
namespace kg {
uint64_t multi_gpu_search(const search_params_t *, const char *, size_t, int, match_result_t *, int *st)
{ if (st) *st = 2; fail("multi-GPU search_buffer not built yet"); return 0; }
}
