// The metafast.sh driver (metafast_amd/cli/metafast_main.cpp) linked WITHOUT the HIP library, for the sanitizer build of its host code
// (tests/test_host_sanitized_cpu.py): option parsing, in.properties / out.properties, the view / bin2fasta readers of .kmers.bin and
// components.bin, the matrix and .vec parsers of dist-matrix-calculator / heatmap-maker run as they are; every entry point that
// needs the GPU fails with a message (the product has no CPU path, and this is not one).  Test infrastructure.
#include "../../include/metafast_hip.h"
#include <stdio.h>
static const char *NO = "this is the sanitizer build of the driver: no GPU entry points";
extern "C" {
const char *mf_last_error(void) { return NO; }
const char *mf_version(void) { return "sanitizer build (no GPU)"; }
int mf_ctx_create(int, int, mf_ctx **out) { if (out) *out = nullptr; return MF_ERR; }
void mf_ctx_destroy(mf_ctx *) {}
int mf_ctx_set_option(mf_ctx *, const char *, int64_t) { return MF_ERR; }
int mf_ctx_synchronize(mf_ctx *) { return MF_ERR; }
int mf_count_reads(mf_ctx *, const char *const *, int, int, int, mf_table **) { return MF_ERR; }
int mf_count_reads_above(mf_ctx *, const char *const *, int, int, int, int, mf_table **, uint64_t *) { return MF_ERR; }
void mf_table_destroy(mf_table *) {}
int mf_table_stats(const mf_table *, uint64_t *, uint64_t *) { return MF_ERR; }
int mf_table_export(const mf_table *, int, uint64_t *, uint16_t *, uint64_t, uint64_t *) { return MF_ERR; }
int mf_table_load_kmers(mf_ctx *, const char *const *, int, int, int, mf_table **) { return MF_ERR; }
int mf_table_write_kmers(const mf_table *, int, const char *, const char *, uint64_t *) { return MF_ERR; }
int mf_table_write_kmers_filtered(const mf_table *, int, mf_table *, int, const char *, uint64_t *) { return MF_ERR; }
int mf_build_unitigs(mf_ctx *, mf_table *, int, int, int, const char *, const char *, uint64_t *) { return MF_ERR; }
int mf_cut_components(mf_ctx *, mf_table *, int, int, int, const char *, const char *, uint64_t *) { return MF_ERR; }
int mf_features(mf_ctx *, const char *, const char *, int, int, const char *, const char *) { return MF_ERR; }
int mf_features_reads(mf_ctx *, const char *, const char *const *, int, int, int, const char *, const char *) { return MF_ERR; }
int mf_features_selected(mf_ctx *, const char *, const char *, int, int, mf_table *, const char *, const char *) { return MF_ERR; }
int mf_features_reads_selected(mf_ctx *, const char *, const char *const *, int, int, int, mf_table *, const char *, const char *) { return MF_ERR; }
int mf_device_count(void) { return 2; }             /* (two entries: the driver's per-device workers run, and stop at mf_ctx_create) */
int mf_ctx_bind_thread(mf_ctx *) { return MF_ERR; }
int mf_device_memory(int, uint64_t *t) { if (t) *t = (uint64_t)288 << 30; return MF_OK; }
int mf_bray_curtis(const int64_t *, int, int, double *) { return MF_ERR; }
int mf_comm_create_local(mf_ctx *const *, int, mf_comm **) { return MF_ERR; }
void mf_comm_destroy(mf_comm *) {}
int mf_cut_components_sharded_files(mf_comm *, const char *const *, int, int, int, int, int, const char *, const char *, uint64_t *) { return MF_ERR; }
}
