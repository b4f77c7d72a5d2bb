// msq_pileup_stream.hip -- streaming bit-plane pileup kernel (fast path).  Placeholder until the kernel lands:
// reports "not handled" so msq_pileup() uses the simple kernel.
#include "msq_internal.h"

int msq_pileup_stream_launch(msq_ctx*, const msq_packed_reads&, const msq_qv_thresholds&, int32_t, int32_t, uint32_t*,
                             uint32_t*, bool* handled)
{
    *handled = false;
    return MSQ_OK;
}
