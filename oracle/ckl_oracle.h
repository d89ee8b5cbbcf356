/* TEST INFRASTRUCTURE ONLY — plain-C CPU restatement of the crackle
 * compress()/decompress() path (seung-lab/crackle, src/crackle.hpp).
 *
 * This is the parity checker ("oracle") for the HIP product in crackle_amd/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Parity status: PINNED — byte-identical to the compiled reference
 * (oracle/_ref) over the golden grid in tests/golden/ (see tests/gen_golden.py)
 * and to the reference's known-answer vectors.  Pin encoding
 * (hash-container iteration order and all) is restated in ckl_oracle_pins.inc.
 */
#ifndef CKL_ORACLE_H
#define CKL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ckl_oracle_last_error(void);
void ckl_oracle_free(void* p);

/* mirrors crackle::compress<LABEL> (src/crackle.hpp:220-257); labels x-fastest */
int ckl_oracle_compress(
	const void* labels, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	uint64_t parallel, unsigned char** out, uint64_t* out_len);

/* compress with the whole-volume decisions imposed (sharded-encode tests) */
int ckl_oracle_compress_ex(
	const void* labels, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	uint64_t parallel,
	int force_crack_format, int force_label_format, int force_stored_width, const uint8_t* forced_model,
	unsigned char** out, uint64_t* out_len);
int ckl_oracle_stats(
	const void* labels, int dtype_bytes, int64_t sx, int64_t sy, int64_t sz,
	uint64_t* max_label, uint64_t* pixel_pairs, uint64_t* first_voxel, uint64_t* last_voxel);
int ckl_oracle_markov_hist(
	const void* labels, int dtype_bytes, int64_t sx, int64_t sy, int64_t sz,
	int crack_format, uint64_t order, uint32_t* hist);

/* mirrors crackle::decompress<LABEL,OUT> (src/crackle.hpp:503-663) */
int ckl_oracle_decompress(
	const unsigned char* buf, uint64_t n, void* out,
	int64_t z_start, int64_t z_end, uint64_t parallel,
	int has_label, uint64_t label);

/* mirrors crackle::cc3d::connected_components (src/cc3d.hpp:371-400) */
int ckl_oracle_connected_components(
	const void* labels, int dtype_bytes,
	int64_t sx, int64_t sy, int64_t sz,
	uint32_t* cc_out, uint64_t* per_slice, uint64_t* N);

/* mirrors crackle::crack_code_to_vcg (src/crackle.hpp:414-425) for slice z */
int ckl_oracle_slice_vcg(
	const unsigned char* buf, uint64_t n, int64_t z, uint8_t* vcg_out);

/* mirrors crackle::crc::crc32c (src/crc.hpp:51-57) */
uint32_t ckl_oracle_crc32c(const uint8_t* data, uint64_t n);

/* operations::voxel_connectivity_graph (src/operations.hpp:667-826), whole volume; vcg: sx*sy*sz bytes */
int ckl_oracle_voxel_connectivity_graph(
	const unsigned char* buf, uint64_t n, int connectivity, uint64_t parallel, uint8_t* vcg);

/* reencode_with_markov_order (src/crackle.hpp:858-984); *out is released with ckl_oracle_free */
int ckl_oracle_reencode(
	const unsigned char* buf, uint64_t n, int markov_order, uint64_t parallel,
	unsigned char** out, uint64_t* out_len);

/* operations::array_equal (src/operations.hpp:1039-1184) */
int ckl_oracle_array_equal(const unsigned char* buf1, uint64_t n1, const unsigned char* buf2, uint64_t n2, uint64_t parallel, int* equal);

/* operations::point_cloud (src/operations.hpp:183-262) with dual_graph::extract_contours
 * (src/dual_graph.hpp:133-275), parallel = 1: labels ascending, offsets in points, (x, y, z) uint16 */
int ckl_oracle_point_cloud(
	const unsigned char* buf, uint64_t n, int64_t z_start, int64_t z_end,
	const uint64_t* labels, uint64_t n_labels, int has_labels, int skip_background,
	uint64_t** labels_out, uint64_t** offsets_out, uint16_t** points_out, uint64_t* n_out);

/* operations::mode_pooling_2x2x1 (src/operations.hpp:1201-1340): per-slice streams one after the other */
int ckl_oracle_mode_pooling(
	const unsigned char* buf, uint64_t n, int64_t z_start, int64_t z_end, uint64_t parallel,
	unsigned char** out, uint64_t* out_len, uint64_t* lens_out, uint64_t* count);

/* operations::voxel_counts / centroids / bounding_boxes (src/operations.hpp:321-665): arrays sorted by
 * label; which 0: 1 x uint64 (count), 1: 3 x double (centroid), 2: 6 x uint32 (box) per label */
int ckl_oracle_label_stats(
	const unsigned char* buf, uint64_t n, int which, int64_t z_start, int64_t z_end, uint64_t parallel,
	uint64_t** labels_out, void** values_out, uint64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif
