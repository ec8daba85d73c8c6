// temporary stubs (replaced as the rows of SURVEY §8 are filled in)
#include "kg_internal.h"
namespace kg {
AcTables *ac_build(const search_params_t &, int) { fail("Aho-Corasick not built yet"); return nullptr; }
void ac_free(AcTables *) {}
int ac_scan(AcTables *, Counters *, Counters *, PostScratch &, int, const uint8_t *, size_t, size_t, size_t,
            size_t, match_position_t *, uint64_t, bool, bool, bool, size_t, hipStream_t, int, hipEvent_t, hipEvent_t,
            krep_gpu_scan_out_t *)
{ return fail("Aho-Corasick not built yet"); }
uint64_t multi_gpu_search(const search_params_t *, const char *, size_t, int, match_result_t *, int *st)
{ if (st) *st = 2; fail("multi-GPU search_buffer not built yet"); return 0; }
}
