// msq_bam_dev.h -- a BAM file resident in HBM (msq_bgzf.hip inflates it, msq_bamdev.hip walks and decodes it).
#pragma once

#include <vector>

#include "msq_internal.h"

constexpr uint64_t MSQ_BAM_DATA_PAD = 256;  // zero bytes behind the inflated stream: header probes may look past its end

struct msq_bam_dev {
    uint8_t* d_data = nullptr;  // the inflated BAM stream + MSQ_BAM_DATA_PAD
    uint64_t data_bytes = 0;
    // msq_bam_index_device: every record's body offset, and the table of all records
    uint64_t* d_rec_off = nullptr;
    msq_bam_record* d_rec = nullptr;
    uint64_t n_records = 0;
    // the records that passed the filters, in file order (set by msq_bam_index_device; what msq_bam_pack_device reads)
    msq_bam_record* d_kept = nullptr;
    uint64_t n_kept = 0;
    std::vector<msq_bam_record> h_kept;  // the same table on the host
    int32_t region_start0 = 0, region_end0 = 0;
};
