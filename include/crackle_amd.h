/* crackle_amd — MI355X-native crackle encode/decode path: C-ABI drop-in boundary.
 *
 * Every entry point replaces (or is the device-resident form of) a function the
 * reference exposes for this path; the reference interface each one stands in
 * for is cited as /root/reference-relative file:line.
 *
 *   reference pybind module   src/fastcrackle.cpp:84-128  (decompress)
 *                             src/fastcrackle.cpp:163-210 (compress)
 *   reference C++ core        src/crackle.hpp:220-257     (crackle::compress<LABEL>)
 *                             src/crackle.hpp:503-663     (crackle::decompress<LABEL,OUT>)
 *   reference C-ABI precedent wasm/crackle_wasm.cc:22-68  (crackle_compress / crackle_decompress)
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the boundary;
 * every function returns a ckl_status and records a thread-local message
 * retrievable with ckl_last_error().  Label volumes are x-fastest ("Fortran
 * order", src/crackle.hpp:219).  All compute runs on the selected HIP device;
 * there is no CPU fallback — without a usable device the calls fail with
 * CKL_ERR_NO_DEVICE.
 */
#ifndef CRACKLE_AMD_H
#define CRACKLE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ckl_status {
	CKL_OK = 0,
	CKL_ERR_FORMAT = 1,     /* not a crackle stream / corrupt header: crackle.FormatError (crackle/headers.py:78-105) */
	CKL_ERR_RUNTIME = 2,    /* std::runtime_error("crackle: ...") in the reference -> RuntimeError */
	CKL_ERR_ARG = 3,        /* bad argument (TypeError/ValueError in the Python shim) */
	CKL_ERR_NO_DEVICE = 4,  /* no usable HIP device: the product path refuses to run on the CPU */
	CKL_ERR_CRC = 5         /* a stored crc32c did not match (the reference swallows these) */
} ckl_status;

/* where a caller-provided buffer lives */
enum { CKL_MEM_HOST = 0, CKL_MEM_DEVICE = 1 };

/* Thread-local message of the last failing call on this thread. */
const char* ckl_last_error(void);

/* ABI version of this library (bumped on any signature change). */
int ckl_abi_version(void);

/* Number of usable HIP devices (0 when none; never fails). */
int ckl_device_count(void);

/* Parsed stream header — replaces crackle::CrackleHeader (src/header.hpp:35-308)
 * and crackle/headers.py:78-124 for callers that need to size outputs. */
typedef struct ckl_header_info {
	uint32_t format_version;
	uint32_t label_format;       /* 0 FLAT, 2 PINS_VARIABLE_WIDTH */
	uint32_t crack_format;       /* 0 IMPERMISSIBLE, 1 PERMISSIBLE */
	uint32_t is_signed;
	uint32_t data_width;
	uint32_t stored_data_width;
	uint32_t sx, sy, sz;
	uint32_t fortran_order;
	uint32_t markov_model_order;
	uint32_t is_sorted;
	uint64_t num_label_bytes;
	uint64_t header_bytes;
} ckl_header_info;

/* Validates magic/version/crc8 exactly as CrackleHeader::assign_from_buffer
 * (src/header.hpp:98-150).  CKL_ERR_FORMAT on failure. */
int ckl_header_info_from_bytes(const uint8_t* buf, uint64_t n, ckl_header_info* out);

/* ---- one-shot entry points ------------------------------------------------ */

/* Replaces fastcrackle.compress (src/fastcrackle.cpp:163-210) /
 * crackle::compress<LABEL> (src/crackle.hpp:220-257).
 *   labels        x-fastest volume of sx*sy*sz elements, dtype_bytes in {1,2,4,8}
 *   labels_mem    CKL_MEM_HOST or CKL_MEM_DEVICE (pointer valid on `device`)
 *   is_signed     must be 0 (crackle/codec.py:720-721 rejects signed input)
 *   allow_pins .. manual_bgcolor   same meaning as the reference's arguments, with two refusals
 *                 (CKL_ERR_ARG): optimize_pins (allow_pins = 2, find_optimal_pins: out of scope) and
 *                 markov_model_order 14 and 15 — the header's four bits allow them
 *                 (src/header.hpp:126,223), but their 4^N x 4 histogram (4 and 16 GiB of counters) is
 *                 more than this encoder keeps on the device; the decoder takes every order
 *   device        HIP device ordinal
 *   out/out_len   library-owned host buffer holding the .ckl bytes; release with ckl_free (only)
 */
int ckl_compress(
	const void* labels, int labels_mem, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_model_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	int device, uint8_t** out, uint64_t* out_len);

void ckl_free(void* p);

/* Replaces fastcrackle.decompress (src/fastcrackle.cpp:84-128) /
 * crackle::decompress<LABEL,OUT> (src/crackle.hpp:503-663).
 *   out            sx*sy*(z_end-z_start) elements of data_width bytes (1 byte each
 *                  when has_label), written x-fastest for fortran_order streams and
 *                  transposed (z fastest) otherwise, as crackle.hpp:617-656 does
 *   out_mem        CKL_MEM_HOST or CKL_MEM_DEVICE
 *   z_start,z_end  slice range; z_end < 0 means sz (clamped like crackle.hpp:527-537)
 */
int ckl_decompress(
	const uint8_t* buf, uint64_t n,
	void* out, uint64_t out_capacity_bytes, int out_mem,
	int64_t z_start, int64_t z_end,
	int has_label, uint64_t label,
	int device);

/* ---- resident sessions (inputs already in HBM; what bench.py times) -------- */

typedef struct ckl_decoder ckl_decoder;

/* Parses the stream on the host (header, z-index + crc32c, label section layout —
 * crackle.hpp:262-336, labels.hpp:424-451), uploads it to HBM and allocates the
 * scratch for slices [z_start, z_end). */
int ckl_decoder_create(
	const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end,
	int device, ckl_decoder** out);
/* The same for a stream that is resident in HBM already (the encoder's ckl_encoder_device_stream, or a
 * caller that keeps its streams on the device): `stream_device` is used in place — it stays the
 * caller's and must outlive the session — and nothing is uploaded.  The host reads back what
 * crackle::decompress parses before its slice loop (header, z-index, src/crackle.hpp:262-313; label
 * section head, src/labels.hpp:424-451; markov model; crc tail) in three small copies, never the
 * crack codes.  Ordered after the device's default stream. */
int ckl_decoder_create_device(
	const uint8_t* stream_device, uint64_t n, int64_t z_start, int64_t z_end,
	int device, ckl_decoder** out);
/* Runs the device pipeline into a DEVICE output buffer and waits for it. */
int ckl_decoder_run(ckl_decoder* d, void* out_device, uint64_t out_capacity_bytes, int has_label, uint64_t label);
/* Runs only the crack decoder of the session: the two crack bit planes of every slice of the
 * range (crack_code_to_vcg, src/crackle.hpp:414-425, reduced to the two bits per pixel the
 * component labelling reads), left in HBM and owned by the decoder.  Plane V bit (x,y): crack
 * between pixels (x-1,y)|(x,y); plane H bit (x,y): between (x,y-1)|(x,y); rows of *row_words
 * 32-pixel words, *plane_words words per slice, slices consecutive. */
int ckl_decoder_crack_planes(
	ckl_decoder* d, const uint32_t** plane_v_device, const uint32_t** plane_h_device,
	uint32_t* row_words, uint64_t* plane_words);
/* Integrity check of the decoder's z-range without producing the volume (the per-slice part of
 * crackle.check, crackle/codec.py:900-948, which decodes slice by slice and notes the failures):
 * the pipeline runs up to the component ids; slice_errors[i] receives 0 or a combination of
 * CKL_SLICE_* bits for slice z_start + i. */
#define CKL_SLICE_BAD_CODE 0x7u        /* crack code malformed: index, range or capacity */
#define CKL_SLICE_BAD_COMPONENTS 0x8u  /* component count differs from the label section */
#define CKL_SLICE_BAD_CRC 0x10u        /* crc32c of the component image differs from the stored one */
int ckl_decoder_check(ckl_decoder* d, uint32_t* slice_errors, uint64_t capacity);

/* Replaces crackle::operations::voxel_connectivity_graph (src/operations.hpp:667-826, bound at
 * src/fastcrackle.cpp:538-565): one byte per voxel of the decoder's z-range, x fastest, bit0 +x,
 * bit1 -x, bit2 +y, bit3 -y (from the crack planes), and for connectivity 6 bit4 +z / bit5 -z
 * where neighbouring slices carry the same label (first slice -z and last slice +z always set).
 * ckl_decoder_vcg writes into a DEVICE buffer; ckl_voxel_connectivity_graph is the one-shot
 * form over the whole volume with a HOST buffer of sx*sy*sz bytes, ckl_voxel_connectivity_graph_range
 * the same for slices [z_start, z_end) (z_end < 0: to the end), like the reference's binding. */
int ckl_decoder_vcg(ckl_decoder* d, uint8_t* out_device, uint64_t out_capacity_bytes, int connectivity);
int ckl_voxel_connectivity_graph(const uint8_t* buf, uint64_t n, int connectivity, int device, uint8_t* out_host, uint64_t out_capacity_bytes);
int ckl_voxel_connectivity_graph_range(const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, int connectivity, int device, uint8_t* out_host, uint64_t out_capacity_bytes);

/* Replaces crackle::operations::array_equal (src/operations.hpp:1039-1184, bound at
 * src/fastcrackle.cpp:594-618): *equal = 1 when both streams have the same shape, the same number
 * of components in every slice and label_map1[components1] == label_map1[components2] everywhere
 * — the FIRST stream's component -> label table on both sides, as the reference has it (:1163-1164);
 * crackle/operations.py:976-994 compares the label sets before it calls this. */
int ckl_array_equal(const uint8_t* buf1, uint64_t n1, const uint8_t* buf2, uint64_t n2, int device, int* equal);

/* Replaces crackle::operations::mode_pooling_2x2x1 (src/operations.hpp:1201-1304, bound at
 * src/fastcrackle.cpp:620-639): slices [z_start, z_end) are decoded, pooled 2 x 2 in x and y by the
 * reference's rule (a == b ? a : a == c ? a : b == c ? b : d) and every pooled slice is encoded as
 * a one-slice stream of its own.  *out holds the streams one after the other (release with
 * ckl_free), *lengths their `*count` lengths (release with ckl_free as well). */
int ckl_mode_pooling_2x2x1(
	const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, int device,
	uint8_t** out, uint64_t* out_len, uint64_t** lengths, uint64_t* count);

/* Per-label statistics of the decoder's z-range without materialising the volume:
 * replaces crackle::operations::voxel_counts / centroids / bounding_boxes
 * (src/operations.hpp:321-618, bound by src/fastcrackle.cpp:346-420).  The pipeline runs up
 * to the run labels; a kernel over the runs accumulates, per label of the stream's label
 * table (flat: the unique list; pins: the unique list and the background color), the voxel
 * count, the sums of the x, y and z coordinates (centroid = sums / count) and the inclusive
 * box [xmin ymin zmin xmax ymax zmax] (z in whole-volume coordinates; a label absent from the
 * range keeps the reference's initial box: mins 0xFFFFFFFF, maxes 0).  The reference seeds its
 * box map from the unique list only (:561-567): the background color of a pin stream, when it is
 * not in that list, is default-constructed on first use, so its three minima are 0; when it is
 * also absent from the range it has no entry in the reference's map and is reported here with
 * count 0 and an all-zero box (callers drop rows with count == 0 and boxes[0] == 0).
 * Host outputs, any of which may be NULL: labels[capacity] (values as the decoder paints
 * them, sign-extended to 64 bits, ascending as unsigned), counts[capacity],
 * sums[3*capacity], boxes[6*capacity].  *n_labels receives the table size; CKL_ERR_ARG
 * with *n_labels set when capacity is too small. */
int ckl_decoder_label_stats(
	ckl_decoder* d, uint64_t capacity, uint64_t* labels, uint64_t* counts,
	uint64_t* sums, uint32_t* boxes, uint64_t* n_labels);
/* Elapsed device time of the last run, from HIP events recorded on the library's
 * own stream around (a) the whole pipeline and (b) the dominant kernel. */
int ckl_decoder_last_timing(const ckl_decoder* d, float* pipeline_ms, float* dominant_kernel_ms);
/* Name and elapsed milliseconds of stage `index` (0-based, in launch order) of the last
 * run; CKL_ERR_ARG past the last stage.  `*name` points to a static string. */
int ckl_decoder_stage_timing(const ckl_decoder* d, int index, const char** name, float* ms);
/* on = 0: the following runs record HIP events only around the whole pipeline (ckl_decoder_last_timing's
 * pipeline_ms stays valid, the per-stage table is empty); on = 1 (the default): also between the kernels.
 * The events between the kernels cost a few microseconds of device time per run.  (No reference counterpart:
 * the reference has no device.) */
int ckl_decoder_set_stage_events(ckl_decoder* d, int on);
void ckl_decoder_destroy(ckl_decoder* d);

typedef struct ckl_encoder ckl_encoder;

/* Allocates device scratch for volumes up to sx*sy*sz of dtype_bytes. */
int ckl_encoder_create(int64_t sx, int64_t sy, int64_t sz, int dtype_bytes, int device, ckl_encoder** out);

/* Optional overrides used when z-slabs of one volume are encoded on several GPUs and
 * merged afterwards: the format decisions that the reference
 * takes from whole-volume reductions (crackle.hpp:48-64, 233-235) are imposed. */
typedef struct ckl_encode_overrides {
	int32_t force_crack_format;      /* -1: decide from this volume; else 0/1 */
	int32_t force_label_format;      /* -1: decide; else 0 (FLAT) / 2 (PINS) */
	int32_t force_stored_width;      /* 0: decide; else 1/2/4/8 */
	int32_t has_model;               /* 1: use `model` (4^order rows x 4 symbol->rank bytes) instead of this volume's statistics */
	const uint8_t* model;
	/* Optional: called once per run, while the crack trail is still executing, with the slab's
	 * distinct labels (host memory; in no particular order when they come straight from the hash
	 * pass: the caller sorts the union anyway).  It returns the sorted unique labels of ALL slabs
	 * (a superset; host memory that stays valid until the run returns); the flat label section
	 * is then written against that list, so that the slabs' sections concatenate without
	 * re-keying.  Return non-zero to abort the run.  NULL: the slab's own list is used. */
	int (*merge_unique)(void* ctx, const uint64_t* local_distinct, uint64_t n_local,
	                    const uint64_t** merged_sorted, uint64_t* n_merged);
	void* merge_ctx;
} ckl_encode_overrides;

/* Encodes a DEVICE-resident volume; result is a library-owned (pinned, cached) host buffer: release with ckl_free. */
int ckl_encoder_run(
	ckl_encoder* e, const void* labels_device,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_model_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	const ckl_encode_overrides* overrides,
	uint8_t** out, uint64_t* out_len);

/* Sharded encoders place every slab's crack codes at an offset of a shared buffer that is only
 * known once all slabs are encoded.  With defer != 0 the following runs leave the crack codes in
 * HBM (the code region of the returned stream is then unspecified; everything else is as usual)
 * and ckl_encoder_codes_to_host copies them, contiguous in slice order, to where they belong:
 * one transfer over the GPU's own link instead of a transfer and a host copy. */
int ckl_encoder_defer_codes(ckl_encoder* e, int defer);
/* With keep != 0 every following run also leaves the whole stream, byte for byte what it returns on
 * the host, in HBM: the encode -> decode round trip of a device-resident volume then never re-uploads
 * its own bytes.  ckl_encoder_device_stream hands out pointer and length of the last run's stream; the
 * buffer is the session's and valid until its next run or its destruction. */
int ckl_encoder_keep_device_stream(ckl_encoder* e, int keep);
/* With on != 0 (and ckl_encoder_keep_device_stream), ckl_encoder_run returns as soon as the stream is complete
 * in HBM: header, z-index and the slices' crcs are in the returned host buffer, its crack codes (the bulk) and —
 * for flat labels — the label section with its crc32c (taken on the device) are still crossing PCIe on a stream
 * of their own.  ckl_encoder_host_wait blocks until they have arrived; the
 * host bytes must not be read, and the buffer not be freed, before.  In between the caller may decode from
 * ckl_encoder_device_stream or do anything else on the device; the encoder's next run waits by itself.
 * With ckl_encoder_defer_codes the same holds for ckl_encoder_codes_to_host: it starts the copy and returns.
 * (The reference returns host bytes from a synchronous call, src/crackle.hpp:220-257: that is the default.) */
int ckl_encoder_async_host_copy(ckl_encoder* e, int on);
int ckl_encoder_host_wait(ckl_encoder* e);
int ckl_encoder_device_stream(const ckl_encoder* e, const uint8_t** stream_device, uint64_t* n_bytes);
int ckl_encoder_codes_to_host(ckl_encoder* e, uint8_t* dst_host, uint64_t capacity, uint64_t* n_bytes);
/* Page-locks / releases a host range for device transfers (e.g. a shared mapping). */
int ckl_host_register(void* p, uint64_t bytes);
int ckl_host_unregister(void* p);

/* Whole-volume reductions of the encode path, exposed so that sharded encoders can
 * all-reduce them (lib.hpp:224-256): max label, count of equal linear neighbours
 * inside this slab, and the slab's first and last voxel (for the pair that straddles
 * a slab boundary).  labels_device is a DEVICE pointer. */
int ckl_encoder_stats(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	uint64_t* max_label, uint64_t* pixel_pairs, uint64_t* first_voxel, uint64_t* last_voxel);

/* Order-N context histogram of this slab's difference-coded crack moves
 * (markov.hpp:193-220): hist must hold 4^order * 4 uint32.  Must follow a
 * ckl_encoder_run-compatible crack pass; see ckl_encoder_run docs in DESIGN.md. */
int ckl_encoder_markov_stats(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	int crack_format, uint64_t markov_model_order, uint32_t* hist);

/* Per-voxel component ids of a DEVICE-resident slab, numbered continuously over its slices
 * from id_base (cc3d::connected_components, src/cc3d.hpp:371-400: 4-connected per slice, ids
 * in first-raster-pixel order), copied to host memory (cc_host: sx*sy*sz uint32, x fastest;
 * ncomp_host: sz counts).  The sharded pin encoder (pins::compute, src/pins.hpp:348-403,
 * needs the whole volume's ids on the host that runs the cover solver) calls it per slab. */
int ckl_encoder_components(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	uint32_t id_base, uint32_t* cc_host, uint32_t* ncomp_host);

/* As ckl_encoder_components, the ids staying on the device (cc_device: sx*sy*sz uint32). */
int ckl_encoder_components_device(
	ckl_encoder* e, const void* labels_device, int64_t sx, int64_t sy, int64_t sz,
	uint32_t id_base, uint32_t* cc_device, uint32_t* ncomp_host);

/* The pin label section (pins::compute, src/pins.hpp:348-403 + labels::encode_condensed_pins,
 * src/labels.hpp:157-344) of a WHOLE volume whose labels (the session's dtype) and global
 * component ids are resident on the session's device; ncomp_host holds the component count of
 * each of the sz slices.  What ckl_encoder_run does for allow_pins after labelling components,
 * as a stage of its own for the sharded encoder: rank 0 collects the slabs' labels and ids
 * (ckl_encoder_components_device) in its HBM and calls this once.  Column runs, their dedup and
 * the component -> pin choice run on the device, the ordered cover on the host.  *out is
 * released with ckl_free. */
int ckl_encoder_pin_labels(
	ckl_encoder* e, const void* labels_device, const uint32_t* cc_device,
	int64_t sx, int64_t sy, int64_t sz, const uint32_t* ncomp_host,
	int stored_width, int auto_bgcolor, int64_t manual_bgcolor,
	uint8_t** out, uint64_t* out_len);

/* ---- the pin stage sharded by rows (BASELINE.json configs[4] on several GPUs) ----------------------------------
 * pins::compute (src/pins.hpp:348-403) needs every (x, y) column over ALL slices (extract_columns, :95-163), so a
 * z-slab cannot run it; a slab of ROWS can, because add_pin compares a run only with the label's last pin in the
 * previous column of the same row (:134-160).  The sharded encoder transposes its z-slabs of labels and component ids
 * into row slabs (all-to-all over xGMI) and every rank calls these on rows [y0, y0 + rows) of every slice (x fastest,
 * then its rows, then z).  All arrays are DEVICE memory of n_components entries owned by the caller, who reduces them
 * over its ranks between the calls: unsigned minimum for first_any / first_kept, unsigned maximum for comp_label,
 * best, ze_plus1 and ids.  Keys name columns of the whole volume: (y * sx + x) * sz + z_start.
 *   first   -> first_any (smallest key of a run starting in the component), first_kept (smallest key of a kept run
 *              containing it, << 16 | its depth; all ones: none), comp_label (label of the components seen, else 0)
 *   best    first_kept reduced -> best (1 + largest key of a kept run deeper than the first; 0: none)
 *   extent  first_kept, best reduced -> choice (the run find_suboptimal_pins draws per component, :325-340; the same
 *           on every rank) and ze_plus1 (its last slice + 1 where this rank holds its row, else 0)
 *   ids     choice, ze_plus1 reduced, offsets (exclusive prefix of the runs' lengths, n_components + 1 entries) ->
 *           ids (component ids along the runs of this rank's rows; the caller zeroes the array first)
 *   section (one rank) the reduced arrays -> the pin label section (encode_condensed_pins, src/labels.hpp:192-344),
 *           released with ckl_free.  ncomp_host: component count of every slice. */
int ckl_pins_rows_first(ckl_encoder* e, const void* labels_rows, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0,
	uint64_t n_components, uint64_t* first_any, uint64_t* first_kept, uint64_t* comp_label);
int ckl_pins_rows_best(ckl_encoder* e, const void* labels_rows, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0,
	uint64_t n_components, const uint64_t* first_kept, uint64_t* best);
int ckl_pins_rows_extent(ckl_encoder* e, const void* labels_rows, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0,
	uint64_t n_components, const uint64_t* first_kept, const uint64_t* best, uint64_t* choice, uint32_t* ze_plus1);
int ckl_pins_rows_ids(ckl_encoder* e, const uint32_t* cc_rows, int64_t sx, int64_t rows, int64_t sz, int64_t y0,
	uint64_t n_components, const uint64_t* choice, const uint32_t* ze_plus1, const uint64_t* offsets, uint32_t* ids);
int ckl_pins_rows_section(ckl_encoder* e, int64_t sx, int64_t sy, int64_t sz, uint64_t n_components, const uint32_t* ncomp_host,
	const uint64_t* comp_label, const uint64_t* first_any, const uint64_t* choice, const uint32_t* ze_plus1, const uint64_t* offsets, const uint32_t* ids,
	int stored_width, int auto_bgcolor, int64_t manual_bgcolor, uint8_t** out, uint64_t* out_len);

int ckl_encoder_last_timing(const ckl_encoder* e, float* pipeline_ms, float* dominant_kernel_ms);
/* Which walk the last run's crack trail took (create_crack_codes, src/crackcodes.hpp:374-453): slices walked by
 * the hand-scheduled k_trail_walk loop / by the compiled one (slices with too many nodes or events for its
 * 16-bit fields, or a branch stack beyond its LDS part).  Tests assert the path with it; no reference
 * counterpart. */
int ckl_encoder_walk_paths(ckl_encoder* e, uint32_t* fast_slices, uint32_t* compiled_slices);
/* Diagnostic: the serial crack trail of the last ckl_encoder_run, by kind of step.  counts[5 * z + k] for slice z:
 * k = 0 steps along the only remaining edge of a node, 1 steps that push a branch and pick the lowest edge
 * (create_crack_codes' one real decision, src/crackcodes.hpp:399-433), 2 dead ends (returns to the branch
 * stack), 3 chain ends, 4 the longest run of steps that take no decision.  max_slices: capacity of counts / 5. */
int ckl_encoder_walk_step_kinds(ckl_encoder* e, uint32_t* counts, uint32_t max_slices, uint32_t* n_slices);
void ckl_encoder_destroy(ckl_encoder* e);

/* ---- host-side stream surgery used by the sharded encoder ------------------ */

/* Concatenates FLAT-label streams of consecutive z-slabs (same sx, sy, dtype, crack
 * format, stored width, markov order+model) into the stream the reference would have
 * produced for the whole volume — native form of crackle.operations.zstack
 * (crackle/operations.py:424-548), host only, no device needed. */
int ckl_zstack(
	const uint8_t* const* bufs, const uint64_t* lens, uint64_t count,
	uint8_t** out, uint64_t* out_len);

/* Host stage of the pin label encoder: candidate pins, greedy cover and the condensed
 * pin section — pins::compute with the fast solver (src/pins.hpp:95-198, 300-403) +
 * labels::encode_condensed_pins (src/labels.hpp:157-344).  `labels` (dtype_bytes wide)
 * and `cc` (global component id of every voxel, as crackle::cc3d::connected_components
 * numbers them, src/cc3d.hpp:371-400) are HOST pointers in Fortran order; `ncomp` holds
 * the component count of each of the sz slices.  ckl_encoder_run and ckl_encoder_pin_labels
 * find the candidate pins on the device and share the ordered cover with this function, which
 * finds them with the reference's own loops; it is exported so that the order-sensitive host
 * logic can be tested without a GPU.  *out is released with ckl_free. */
int ckl_pin_labels_host(
	const void* labels, int dtype_bytes, const uint32_t* cc,
	int64_t sx, int64_t sy, int64_t sz, const uint32_t* ncomp,
	int stored_width, int auto_bgcolor, int64_t manual_bgcolor,
	uint8_t** out, uint64_t* out_len);

/* Replaces crackle::operations::point_cloud (src/operations.hpp:183-262, bound as
 * fastcrackle.point_cloud, src/fastcrackle.cpp:315-345) with dual_graph::extract_contours
 * (src/dual_graph.hpp:133-275): the boundary contours of every 2D component of slices
 * [z_start, z_end) (clamped like operations::get_szr; an empty range is an error), as (x, y, z)
 * uint16 triples per label.  `labels` (when has_labels) restricts the output to those labels,
 * skip_background drops label 0.  The labels that own points come back ascending (the reference
 * returns an unordered_map), offsets_out[i] .. offsets_out[i+1] are label i's points in
 * points_out (3 uint16 each), in the order the reference appends them with parallel = 1: slices
 * ascending, components in index order, each component's contours merged as the reference merges
 * them.  The crack codes are decoded and the contours traced on the device (one wavefront per
 * slice).  Release the three arrays with ckl_free. */
int ckl_point_cloud(
	const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end,
	const uint64_t* labels, uint64_t n_labels, int has_labels, int skip_background, int device,
	uint64_t** labels_out, uint64_t** offsets_out, uint16_t** points_out, uint64_t* n_out);

/* Replaces crackle::reencode_with_markov_order (src/crackle.hpp:858-984, bound as
 * fastcrackle.reencode_markov, src/fastcrackle.cpp:212-230): the stream with its crack codes
 * stored under another markov model order.  The crack decoder rasterises the codes on the device
 * and the encoder's crack trail writes them out again; label section and crcs are copied.
 * Streams written by the reference encoder or by this library come out byte for byte as the
 * reference's reencode gives them.  *out (pinned host memory) is released with ckl_free. */
int ckl_reencode_markov(const uint8_t* buf, uint64_t n, int markov_model_order, int device, uint8_t** out, uint64_t* out_len);

/* Native form of crackle.operations.zsplit's range helper (crackle/operations.py:550-623):
 * the stream of slices [z_start, z_end) of a FLAT stream, without decoding (host only).
 * *out is released with ckl_free. */
int ckl_zsplit(const uint8_t* buf, uint64_t n, int64_t z_start, int64_t z_end, uint8_t** out, uint64_t* out_len);

/* crc32c (Castagnoli; src/crc.hpp:51-57) of a host buffer — exported for tests. */
uint32_t ckl_crc32c(const uint8_t* data, uint64_t n);
/* crc32c(A || B) from crc32c(A), crc32c(B) and the byte length of B: lets the ranks of a sharded
 * encode checksum their own parts of the merged label section. */
uint32_t ckl_crc32c_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

#ifdef __cplusplus
}
#endif
#endif /* CRACKLE_AMD_H */
