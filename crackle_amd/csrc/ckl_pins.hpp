// Pin label encoding: what the device hands to the host cover (ckl_pins.hip).
//
// pins::compute (src/pins.hpp:348-403) walks the label volume and the component-id volume
// three times on one core (extract_columns, compute_multiverse, the component -> pins index
// of find_suboptimal_pins).  Here those walks are device passes (ckl_pins_dev.hpp) and only
// per-component facts cross PCIe:
//
//   * the label of every component (component ids ascend in first-appearance order, which is
//     the insertion order of compute_multiverse, src/pins.hpp:165-198);
//   * the first column run of every component in extract_columns' traversal order
//     (y, x, z_start) — the order the labels enter `pinsets`;
//   * the pin find_suboptimal_pins takes when the component is drawn from the universe
//     (src/pins.hpp:325-340).  Candidate pins are never removed, only components are, so
//     that choice is a function of the component alone: with the candidate pins containing
//     it in vector order p0, p1, ..., it is the LAST p_i deeper than p0, or p0
//     (`max_depth` is never updated);
//   * for every pin so chosen its column, z-range and the component ids along it.
#pragma once
#include "ckl_common.hpp"

#include <functional>
#include <memory>
#include <vector>

namespace ckl {

constexpr uint64_t kPinNoKey = ~0ull;
constexpr uint32_t kPinNone = 0xFFFFFFFFu;

// order of extract_columns (src/pins.hpp:126-163): rows, then columns, then z
inline uint64_t pin_key(uint64_t x, uint64_t y, uint64_t z_s, uint64_t sx, uint64_t sz) { return (y * sx + x) * sz + z_s; }

struct PinCandidates {
	std::vector<uint64_t> comp_label;     // [N] label of component c
	std::vector<uint64_t> comp_first;     // [N] smallest pin_key of a column run starting inside c (kPinNoKey: none)
	std::vector<uint64_t> label_value, label_first;   // optional: every label once with the smallest comp_first of its components (else derived from the two above)
	std::vector<uint32_t> comp_pin;       // [N] pin taken when c is drawn (index below; kPinNone cannot happen).  Pins need not be distinct:
	                                      // a pin is taken at most once, because taking it removes every component that maps to it
	std::vector<uint32_t> pin_x, pin_y, pin_zs, pin_ze;
	std::vector<uint64_t> pin_ids_off;    // [P + 1]
	std::vector<uint32_t> pin_ids;        // component ids along each pin, z ascending
	// The three large arrays may live in memory held by the caller instead (the pinned block the device arrays
	// were copied into): when view_components is set, these are read and the vectors of the same name are not.
	uint64_t view_components = 0;
	const uint64_t* view_comp_label = nullptr;
	const uint64_t* view_pin_ids_off = nullptr;
	const uint32_t* view_pin_ids = nullptr;
	// likewise the per-pin arrays of a caller whose pins are the components' own choices (P = N): comp_pin, pin_x,
	// pin_y, pin_zs, pin_ze, each view_components entries
	const uint32_t* view_comp_pin = nullptr;
	const uint32_t* view_pin_x = nullptr;
	const uint32_t* view_pin_y = nullptr;
	const uint32_t* view_pin_zs = nullptr;
	const uint32_t* view_pin_ze = nullptr;
};

// worker threads for host loops over components / labels: body(lo, hi) on disjoint ranges, errors rethrown
void host_parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& body, size_t max_threads = 64);

// The order-sensitive part (cover, background colour, section bytes) on the host.
std::vector<uint8_t> pins_cover_host(
	const PinCandidates& pc, int64_t sx, int64_t sy, int64_t sz,
	const std::vector<uint32_t>& ncomp, uint64_t n_total,
	int index_width, int stored_width, bool auto_bgcolor, int64_t manual_bgcolor,
	const std::function<void()>& components_ready = std::function<void()>(), const struct PinLabelTable* table = nullptr);
// `pinsets` of pins::compute (src/pins.hpp:126-163) from label_value / label_first alone: a caller that has these
// before the rest builds the table while its device passes still run and hands it to pins_cover_host.
struct PinLabelTable;
std::shared_ptr<const PinLabelTable> pins_label_table_host(const std::vector<uint64_t>& label_value, const std::vector<uint64_t>& label_first);
// components_ready: called once the labels' table (`pinsets`, from label_value / label_first alone) is built and
// before anything per component or per pin is read — a caller whose large arrays are still on their way from the
// device waits for them there (and fills comp_pin, pin_x ... pin_ze, whose sizes must be final beforehand).

// PinCandidates from host volumes with the reference's own loops (add_pin vectors): the
// CPU statement the device passes are tested against, and the sharded codec's whole-volume
// stage (ckl_pin_labels_host).
template <typename LABEL>
PinCandidates pin_candidates_host(const LABEL* labels, const uint32_t* cc, int64_t sx, int64_t sy, int64_t sz, uint64_t n_total);

}  // namespace ckl
