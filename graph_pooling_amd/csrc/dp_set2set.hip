// Set2Set readout (set2set.py:32-57) — placeholder until the persistent kernel lands.
#include "dp_common.h"

namespace dp {

size_t set2set_save_bytes(int B, int n, int d) { return 256; }

void set2set_fwd(Seq& q, const float*, int, const float*, const float*, const float*, const float*, const float*,
                 const float*, float*, int, int, int, void*) {
    if (q.err || q.dry) return;
    set_error("Set2Set HIP kernel not built yet");
    q.err = DP_ERR_UNSUPPORTED;
}
void set2set_bwd(Seq& q, const float*, int, const float*, const float*, const float*, const float*, const float*,
                 const float*, const float*, const float*, float*, int, float*, float*, float*, float*, float*, float*,
                 int, int, int, const void*) {
    if (q.err || q.dry) return;
    set_error("Set2Set HIP kernel not built yet");
    q.err = DP_ERR_UNSUPPORTED;
}

}  // namespace dp
