// shim/object_slots.h -- the drop-in's per-object device state and its leases, free of Eigen and HIP so that the same code
// is built under ThreadSanitizer / AddressSanitizer without a GPU (shim/test_concurrency.cc, `make -C shim tsan`).
//
// One context PER OBJECT (the node loops over the objects of a frame, ObjectPoseCandidateSet.cpp:53-68 per object,
// SceneCfg.cpp:379-402): an object's pair-feature table (5 MB at 18 682 keys: milliseconds to flatten, upload and
// hash), its validation model (Morton sort + upload) and its search model stay resident in ITS context from frame
// to frame, so that alternating objects do not evict each other; what a call uploads is the segment.  An object is
// recognised by its PPFMap (address, size, fingerprint of its end entries); the models are compared by a hash of
// their (centred) coordinates and re-sent only when they changed.  At most kSlots objects, least recently used out.
#pragma once

#include <cstddef>
#include <mutex>

#include "../include/pgp.h"

namespace shimstate {

struct ObjectSlot {
  pgp_ctx* ctx = nullptr;
  const void* map_addr = nullptr;
  size_t map_size = 0;
  unsigned long long map_print = 0;
  bool map_loaded = false;
  unsigned long long model_hash = 0, search_hash = 0;
  unsigned long long stamp = 0;
  std::mutex busy;   // held by the call that is matching this object (ShimState::acquire .. SlotLease)
};

// The state is the PROCESS's, not a thread's (it was thread-local up to round 5): whichever thread brings an object finds
// its context -- a ROS callback on another spinner thread, the std::thread per object the reference's authors left
// commented out around this call (SceneCfg.cpp:377,402-403, ObjectPoseCandidateSet.cpp:64-65), a worker of
// getProbableTransformsSuper4PCSFrame.  Calls for different objects run side by side; two calls for the SAME object take
// turns (the second waits for the first's lease).  Never destroyed: the contexts go with the process (no HIP call from
// an exit handler or a thread-local destructor, which the profiler's tooling does not survive).
struct ShimState {
  static constexpr int kSlots = 16;
  ObjectSlot slot[kSlots];
  std::mutex mu;                      // guards the slots' identity fields and the clock
  std::mutex single_mu;               // held for the length of a call that uses `ctx` / `group` below
  unsigned long long clock = 0;
  pgp_ctx* ctx = nullptr;             // several devices: the single context of older rounds
  pgp_multi* group = nullptr;
  const void* map_addr = nullptr;
  size_t map_size = 0;
  unsigned long long map_print = 0;   // fingerprint of the map's two end entries
  const void* map_ctx = nullptr;      // the context the table was uploaded to

  // (mu held) the slot keyed to this object, if any.  A slot counts as keyed from the moment a call claims it (stamp != 0),
  // before its context exists: a second arrival for an object that is being installed finds the first one's slot and waits
  // for its lease instead of keying a slot of its own.
  ObjectSlot* find(const void* addr, size_t size, unsigned long long print) {
    for (ObjectSlot& o : slot)
      if (o.map_addr == addr && o.map_size == size && o.map_print == print && (o.ctx || o.stamp)) return &o;
    return nullptr;
  }
  // (mu and o.busy held) the slot becomes this object's; its context keeps its allocations
  void rekey(ObjectSlot& o, const void* addr, size_t size, unsigned long long print) {
    o.map_addr = addr;
    o.map_size = size;
    o.map_print = print;
    o.map_loaded = false;
    o.model_hash = o.search_hash = 0;
    o.stamp = ++clock;
  }
  // the object's slot, leased to the caller (slot->busy held): its own from an earlier call, or the least recently used
  // one that nobody is using, re-keyed
  ObjectSlot* acquire(const void* addr, size_t size, unsigned long long print) {
    for (;;) {
      ObjectSlot* pick = nullptr;
      bool mine = false;
      {
        std::lock_guard<std::mutex> lk(mu);
        pick = find(addr, size, print);
        mine = pick != nullptr;
        if (!pick) {
          ObjectSlot* lru = nullptr;
          for (ObjectSlot& o : slot) {
            if (!o.busy.try_lock()) continue;
            if (!lru || o.stamp < lru->stamp) {
              if (lru) lru->busy.unlock();
              lru = &o;
            } else {
              o.busy.unlock();
            }
          }
          if (lru) {
            rekey(*lru, addr, size, print);
            return lru;
          }
          pick = &slot[0];   // every slot is in use: wait for the least recently used one
          for (ObjectSlot& o : slot)
            if (o.stamp < pick->stamp) pick = &o;
        }
      }
      pick->busy.lock();
      {
        std::lock_guard<std::mutex> lk(mu);
        // While this call waited, another call may have brought the SAME object (all slots were busy when both arrived, and
        // each set out to wait for a victim): whoever got its victim first has keyed it.  Up to round 5 the second then
        // keyed ITS victim as well -- two slots, two contexts, two uploads of one object's table until one aged out.
        ObjectSlot* now = find(addr, size, print);
        if (now == pick) {          // this very slot is the object's (it was before the wait, or it became so during it)
          pick->stamp = ++clock;
          return pick;
        }
        if (!now && !mine) {        // waited for a victim, the object still has no slot: it is free now, take it over
          rekey(*pick, addr, size, print);
          return pick;
        }
        // the object's slot was given away while this call waited for it, or the object got ANOTHER slot meanwhile: look again
      }
      pick->busy.unlock();
    }
  }
  ~ShimState() {   // (only a call's own state under PGP_SHIM_NO_CACHE is ever destroyed)
    if (group) pgp_multi_destroy(group);
    if (ctx) pgp_destroy(ctx);
    for (ObjectSlot& o : slot)
      if (o.ctx) pgp_destroy(o.ctx);
  }
};

struct SlotLease {
  ObjectSlot* s = nullptr;
  ~SlotLease() {
    if (s) s->busy.unlock();
  }
};

}  // namespace shimstate
