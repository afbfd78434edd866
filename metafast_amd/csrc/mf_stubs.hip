// TEMPORARY: entry points not implemented yet (replaced file by file).
#include "mf_common.h"
#define NI(name) return mf_set_error(#name ": not implemented yet")
extern "C" int mf_count_reads(mf_ctx *, const char *const *, int, int, int, mf_table **) { NI(mf_count_reads); }
extern "C" int mf_table_write_kmers(const mf_table *, int, const char *, const char *, uint64_t *) { NI(mf_table_write_kmers); }
extern "C" int mf_table_load_kmers(mf_ctx *, const char *const *, int, int, int, mf_table **) { NI(mf_table_load_kmers); }
extern "C" int mf_build_unitigs_device(mf_ctx *, mf_table *, int, int, mf_seqs **) { NI(mf_build_unitigs_device); }
extern "C" void mf_seqs_destroy(mf_seqs *) {}
extern "C" int mf_seqs_stats(const mf_seqs *, uint64_t *, uint64_t *) { NI(mf_seqs_stats); }
extern "C" int mf_seqs_device_view(const mf_seqs *, const void **, const void **, const void **, const void **, const void **, uint64_t *, uint64_t *) { NI(mf_seqs_device_view); }
extern "C" int mf_seqs_export(const mf_seqs *, uint8_t *, uint64_t *, int32_t *, int32_t *, int32_t *) { NI(mf_seqs_export); }
extern "C" int mf_seqs_write_fasta(const mf_seqs *, const char *) { NI(mf_seqs_write_fasta); }
extern "C" int mf_build_unitigs(mf_ctx *, mf_table *, int, int, int, const char *, const char *, uint64_t *) { NI(mf_build_unitigs); }
extern "C" int mf_cut_components_device(mf_ctx *, mf_table *, int, int, mf_comps **) { NI(mf_cut_components_device); }
extern "C" void mf_comps_destroy(mf_comps *) {}
extern "C" int mf_comps_stats(const mf_comps *, uint64_t *, uint64_t *) { NI(mf_comps_stats); }
extern "C" int mf_comps_export(const mf_comps *, uint64_t *, int64_t *, int32_t *, uint64_t *, uint64_t *) { NI(mf_comps_export); }
extern "C" int mf_comps_write(const mf_comps *, const char *, const char *) { NI(mf_comps_write); }
extern "C" int mf_comps_load(mf_ctx *, const char *, mf_comps **) { NI(mf_comps_load); }
extern "C" int mf_cut_components(mf_ctx *, mf_table *, int, int, int, const char *, const char *, uint64_t *) { NI(mf_cut_components); }
extern "C" int mf_features_device(mf_ctx *, mf_comps *, const mf_table *, int, int64_t *, double *) { NI(mf_features_device); }
extern "C" int mf_features(mf_ctx *, const char *, const char *, int, int, const char *, const char *) { NI(mf_features); }
extern "C" int mf_bray_curtis(const int64_t *, int, int, double *) { NI(mf_bray_curtis); }
