// eppk_pick_inst.hip.h — the pick kernels are instantiated in six translation units (one per lane-word type and counter-plane
// count), compiled in parallel by __graft_entry__.build().  A unit defines EPPK_PICK_INST_LW / _NPL / _NAME and includes this
// header after eppk_kernels.hip.h; eppk.hip includes it for the declarations only.
#pragma once

namespace eppk {

struct PickVariant {       // which instantiation a context needs (eppk.hip: pick_kernel_ptr)
  bool fast;               // fused sparse kernel (else generic per-pair kernel)
  bool masked, topk;       // candidate masks; ordered fallbacks (TOPK instantiation of the fast / generic kernel)
  bool big;                // index of 4 GiB and more
  bool has_l, has_p, p_first, gen;
};

const void* pick_kernel_u16_6(const PickVariant& v);
const void* pick_kernel_u16_9(const PickVariant& v);
const void* pick_kernel_u32_6(const PickVariant& v);
const void* pick_kernel_u32_9(const PickVariant& v);
const void* pick_kernel_u64_6(const PickVariant& v);
const void* pick_kernel_u64_9(const PickVariant& v);
// pick_quad_kernel (four requests per wavefront; prefix scorer present, <= 63 blocks, no interpreted tail) and the work-list
// instantiation of the fast kernel that scores what it defers; defined in eppk_pick_quad.hip / eppk_pick_wl.hip
const void* pick_quad_u16(bool has_l, bool p_first, bool masked, bool topk);
const void* pick_quad_u32(bool has_l, bool p_first, bool masked, bool topk);
const void* pick_quad_u64(bool has_l, bool p_first, bool masked, bool topk);
// the one-launch form (every workgroup scores what it deferred itself): one translation unit per (masked, topk) pair,
// eppk_pick_quad_tail[_masked][_topk].hip; eppk_pick_quad_tail.hip also holds the dispatchers
const void* pick_quad_tail_u16(bool has_l, bool p_first, bool masked, bool topk);
const void* pick_quad_tail_u32(bool has_l, bool p_first, bool masked, bool topk);
const void* pick_quad_tail_u64(bool has_l, bool p_first, bool masked, bool topk);
const void* pick_quad_tail_masked(int lw_bytes, bool has_l, bool p_first);
const void* pick_quad_tail_topk(int lw_bytes, bool has_l, bool p_first);
const void* pick_quad_tail_topk_masked(int lw_bytes, bool has_l, bool p_first);
const void* pick_quad_tail_learn(int lw_bytes, bool has_l, bool p_first);          // LEARN instantiations (single picks; eppk_pick_quad_tail_learn[_masked].hip)
const void* pick_quad_tail_learn_masked(int lw_bytes, bool has_l, bool p_first);
const void* pick_resident(int lw_bytes, bool has_l, bool p_first);               // the resident small-batch kernel (eppk_pick_resident.hip)
const void* pick_resident_quad(int lw_bytes, bool has_l, bool p_first);          // ... with pick_quad_kernel's body (eppk_pick_resident_quad.hip)
const void* pick_resident_quad_masked(int lw_bytes, bool has_l, bool p_first);       // ... its variants, a unit each (eppk_pick_resident_quad_*.hip)
const void* pick_resident_quad_topk(int lw_bytes, bool has_l, bool p_first);
const void* pick_resident_quad_topk_masked(int lw_bytes, bool has_l, bool p_first);
const void* pick_resident_quad_variant(int lw_bytes, bool has_l, bool p_first, bool masked, bool topk);
const void* pick_resident_quad_learn(int lw_bytes, bool has_l, bool p_first, bool masked);        // ... followed by the index update (eppk_pick_resident_quad_learn*.hip)
const void* pick_resident_quad_learn_masked(int lw_bytes, bool has_l, bool p_first);
const void* pick_fast_wl_topk_u16(bool has_l, bool p_first, bool big);          // work-list instantiations with ordered fallbacks (eppk_pick_wl_topk.hip)
const void* pick_fast_wl_topk_u32(bool has_l, bool p_first, bool big);
const void* pick_fast_wl_topk_u64(bool has_l, bool p_first, bool big);
const void* pick_fast_wl_topk_masked_u16(bool has_l, bool p_first);               // ... with candidate masks as well (eppk_pick_wl_topk_masked.hip)
const void* pick_fast_wl_topk_masked_u32(bool has_l, bool p_first);
const void* pick_fast_wl_topk_masked_u64(bool has_l, bool p_first);
const void* pick_fast_wl_u16(bool has_l, bool p_first, bool big, bool masked);
const void* pick_fast_wl_u32(bool has_l, bool p_first, bool big, bool masked);
const void* pick_fast_wl_u64(bool has_l, bool p_first, bool big, bool masked);

#ifdef EPPK_PICK_INST_NAME
template <typename LW, int NPL, bool MASKED, bool BIG, bool TOPK>
static const void* fast_kernel_ptr(const PickVariant& v) {
  if (v.gen) {   // interpreted tail (pod-only scorers behind LORA / PREFIX); the order lives in KTail
    if (v.has_l && v.has_p) return (const void*)pick_fast_kernel<LW, NPL, true, true, false, MASKED, BIG, true, TOPK>;
    if (v.has_l) return (const void*)pick_fast_kernel<LW, NPL, true, false, false, MASKED, false, true, TOPK>;
    return (const void*)pick_fast_kernel<LW, NPL, false, true, false, MASKED, BIG, true, TOPK>;     // gen implies LORA or PREFIX
  }
  if (v.has_l && v.has_p) return v.p_first ? (const void*)pick_fast_kernel<LW, NPL, true, true, true, MASKED, BIG, false, TOPK>
                                           : (const void*)pick_fast_kernel<LW, NPL, true, true, false, MASKED, BIG, false, TOPK>;
  if (v.has_l) return (const void*)pick_fast_kernel<LW, NPL, true, false, false, MASKED, false, false, TOPK>;     // no prefix scorer: no index access
  if (v.has_p) return (const void*)pick_fast_kernel<LW, NPL, false, true, false, MASKED, BIG, false, TOPK>;
  return (const void*)pick_fast_kernel<LW, NPL, false, false, false, MASKED, false, false, TOPK>;
}
template <typename LW, int NPL, bool MASKED, bool BIG>
static const void* fast_kernel_ptr(const PickVariant& v) {
  return v.topk ? fast_kernel_ptr<LW, NPL, MASKED, BIG, true>(v) : fast_kernel_ptr<LW, NPL, MASKED, BIG, false>(v);
}

const void* EPPK_PICK_INST_NAME(const PickVariant& v) {
  using LW = EPPK_PICK_INST_LW;
  constexpr int NPL = EPPK_PICK_INST_NPL;
  if (!v.fast) {
    if (v.topk) return v.masked ? (const void*)pick_generic_kernel<LW, NPL, true, (int)EPPK_MAX_TOPK> : (const void*)pick_generic_kernel<LW, NPL, false, (int)EPPK_MAX_TOPK>;
    return v.masked ? (const void*)pick_generic_kernel<LW, NPL, true, 1> : (const void*)pick_generic_kernel<LW, NPL, false, 1>;
  }
  if (v.big) return v.masked ? fast_kernel_ptr<LW, NPL, true, true>(v) : fast_kernel_ptr<LW, NPL, false, true>(v);
  return v.masked ? fast_kernel_ptr<LW, NPL, true, false>(v) : fast_kernel_ptr<LW, NPL, false, false>(v);
}
#endif

}  // namespace eppk

#ifdef EPPK_QUAD_TAIL_UNIT
// (the body of a eppk_pick_quad_tail*.hip unit: EPPK_QUAD_TAIL_UNIT = its function's name, EPPK_QUAD_TAIL_MASKED / _TOPK = its pair)
namespace eppk {
template <typename LW>
static const void* quad_tail_unit_ptr(bool has_l, bool p_first) {
  constexpr bool M = EPPK_QUAD_TAIL_MASKED, T = EPPK_QUAD_TAIL_TOPK;
#ifdef EPPK_QUAD_TAIL_LEARN
  constexpr bool L = true;
#else
  constexpr bool L = false;
#endif
  if (has_l) return p_first ? (const void*)pick_quad_kernel<LW, true, true, M, T, true, L> : (const void*)pick_quad_kernel<LW, true, false, M, T, true, L>;
  return (const void*)pick_quad_kernel<LW, false, false, M, T, true, L>;
}
const void* EPPK_QUAD_TAIL_UNIT(int lw_bytes, bool has_l, bool p_first) {
  return lw_bytes == 2 ? quad_tail_unit_ptr<uint16_t>(has_l, p_first) : lw_bytes == 4 ? quad_tail_unit_ptr<uint32_t>(has_l, p_first) : quad_tail_unit_ptr<uint64_t>(has_l, p_first);
}
}  // namespace eppk
#endif
