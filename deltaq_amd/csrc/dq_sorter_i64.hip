// dq_sorter_i64.hip -- the suffix sorter for 64-bit suffix indices (dq_sorter_impl.h).
#include "dq_sorter_impl.h"

namespace dq {
template int sufsort_host<int64_t>(const uint8_t *, int64_t, int64_t *, int32_t, SortHints);
template int sufsort_dev<int64_t>(const void *, int64_t, void *, int32_t, void *);
template int64_t sufsort_workspace_bytes<int64_t>(int64_t);
}  // namespace dq
