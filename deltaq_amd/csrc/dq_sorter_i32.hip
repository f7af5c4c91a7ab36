// dq_sorter_i32.hip -- the suffix sorter for 32-bit suffix indices (dq_sorter_impl.h).
#include "dq_sorter_impl.h"

namespace dq {
template int sufsort_host<int32_t>(const uint8_t *, int64_t, int32_t *, int32_t, SortHints);
template int sufsort_dev<int32_t>(const void *, int64_t, void *, int32_t, void *);
template int64_t sufsort_workspace_bytes<int32_t>(int64_t);
}  // namespace dq
