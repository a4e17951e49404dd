// megakernel_wg.inl — the persistent megakernel with ONE path pool per workgroup of WAVES waves
// (included by kernels.hip after megakernel.inl, whose helpers, path-slot layout and pass bodies it shares).
//
// Why: with one pool of 128 slots per wave (megakernel.inl) the five work queues are shallow, so a
// shading pass finds only 30-45 paths of one kind for its 64 lanes (profiles: passes_* / slots_*).
// Here WAVES waves share POOLN slots in LDS and hand paths to each other through lock-free per-kind
// queues: deeper queues -> fuller passes, and the per-round census over the status array disappears.
//
// Queues (one per kind: EMPTY, TRAV, TERM, LAMB, METAL, DIEL), all in LDS:
//   push : a wave reserves positions with ONE ds_add on the queue's tail for all its lanes that push
//          (ballot + rank), then each lane stores slot+1 into its position.
//   pop  : lane 0 reserves [head, head+n) with a compare-and-swap on the head (n <= tail - head, so only
//          positions some producer has already reserved), the lanes then read their entries, waiting
//          for a producer that has reserved but not yet stored (entries are 0 until stored), and zero them.
// A slot id is in exactly one queue or owned by one lane, so a queue never holds more than POOLN
// entries and a position is not reused before it has been consumed. Every wait is bounded; running
// out of the bound sets DevCounters::diag[23] and ends the wave (tests fail loudly instead of hanging).

template <int WAVES, int POOLN, bool STATS>
__global__ __launch_bounds__(64 * WAVES, RBRT_MK_WAVES_PER_SIMD) void trace_megakernel_wg(const TraceParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t kQ = kNumStatus;
    uint32_t* const pool = lds;                                        // [kFields][POOLN]
    uint32_t* const qhead = pool + kFields * POOLN;                    // [8]
    uint32_t* const qtail = qhead + 8;                                 // [8]
    uint16_t* const qent = reinterpret_cast<uint16_t*>(qtail + 8);     // [kQ][POOLN]: slot + 1, 0 = not stored yet
    constexpr uint32_t kQentDw = (kQ * uint32_t(POOLN) + 1u) / 2u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* const stacks = qtail + 8 + kQentDw;
    uint32_t* const stack = stacks + wave * P.stack_entries * 64u + lane;
    uint32_t* const sc_base = stacks + uint32_t(WAVES) * P.stack_entries * 64u;
    uint32_t* const gseq = P.gseq + size_t(blockIdx.x) * POOLN * kSeqWords;
#define WPOOL(f, s) pool[(f) * POOLN + (s)]

    // ---- one-time setup by the whole workgroup ----
    for (uint32_t i = threadIdx.x; i < kQ * uint32_t(POOLN); i += 64u * WAVES)
        qent[i] = uint16_t(i < uint32_t(POOLN) ? i + 1u : 0u);  // queue 0 (EMPTY) starts out holding every slot
    if (threadIdx.x < 16u) qhead[threadIdx.x] = (threadIdx.x == 8u + ST_EMPTY) ? uint32_t(POOLN) : 0u;  // heads, then tails
    const uint32_t n_obj = P.n_spheres + P.n_meshes;
    {
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(P.spheres);
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(P.materials);
        const uint32_t* gh = reinterpret_cast<const uint32_t*>(P.meshes);
        uint32_t* dst = sc_base;
        for (uint32_t i = threadIdx.x; i < P.n_spheres * kSphDw; i += 64u * WAVES) dst[i] = gs[i];
        dst += P.n_spheres * kSphDw;
        for (uint32_t i = threadIdx.x; i < n_obj * kMatDw; i += 64u * WAVES) dst[i] = gm[i];
        dst += n_obj * kMatDw;
        for (uint32_t i = threadIdx.x; i < P.n_meshes * kMeshDw; i += 64u * WAVES) dst[i] = gh[i];
    }
    const SceneLds sc = {reinterpret_cast<const float*>(sc_base), sc_base + P.n_spheres * kSphDw,
                         sc_base + P.n_spheres * kSphDw + n_obj * kMatDw};
    __syncthreads();  // the only workgroup-wide barrier; from here on the waves run independently

    bool fatal = false;  // wave-uniform: a bounded wait ran out
    // ---- queue primitives (wave-level; call from wave-uniform control flow) ----
    auto q_depth = [&](uint32_t k) -> uint32_t {
        const uint32_t t = __hip_atomic_load(&qtail[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t h = __hip_atomic_load(&qhead[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t - h;  // may lag by a concurrent pop; only used as a scheduling hint and as an upper bound in q_pop
    };
    // lanes with kq < kQ push `slot` to queue kq
    auto q_push = [&](uint32_t kq, uint32_t slot) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // slot state (LDS + gseq) before the id becomes visible
#pragma unroll
        for (uint32_t k = 0; k < kQ; ++k) {
            const uint64_t mask = __ballot(kq == k);
            if (mask) {
                const uint32_t leader = uint32_t(__builtin_ctzll(mask));
                uint32_t pos = 0;
                if (lane == leader) pos = atomicAdd(&qtail[k], uint32_t(__popcll(mask)));
                pos = uint32_t(__shfl(int(pos), int(leader)));
                if (kq == k) qent[k * uint32_t(POOLN) + (pos + lane_rank(mask)) % uint32_t(POOLN)] = uint16_t(slot + 1u);
            }
        }
    };
    // up to `want` entries of queue k go to lanes [first_lane, first_lane + n); returns n (wave-uniform)
    auto q_pop = [&](uint32_t k, uint32_t want, uint32_t first_lane, uint32_t& slot_out) -> uint32_t {
        uint32_t h = 0, n = 0;
        if (lane == 0 && want != 0) {
            for (int tries = 0; tries < 32; ++tries) {
                const uint32_t hh = __hip_atomic_load(&qhead[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t tt = __hip_atomic_load(&qtail[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t avail = tt - hh;
                const uint32_t nn = want < avail ? want : avail;
                if (nn == 0 || nn > uint32_t(POOLN)) break;
                if (atomicCAS(&qhead[k], hh, hh + nn) == hh) {
                    h = hh;
                    n = nn;
                    break;
                }
            }
        }
        h = uint32_t(__shfl(int(h), 0));
        n = uint32_t(__shfl(int(n), 0));
        bool bad = false;
        if (lane >= first_lane && lane < first_lane + n) {
            volatile uint16_t* e = qent + k * uint32_t(POOLN) + (h + (lane - first_lane)) % uint32_t(POOLN);
            uint32_t v = *e;
            for (uint32_t spin = 0; v == 0u && spin < (1u << 22); ++spin) {
                __builtin_amdgcn_s_sleep(1);
                v = *e;
            }
            bad = v == 0u;
            *e = 0;
            slot_out = v - 1u;
        }
        if (__any(bad)) {
            fatal = true;
            n = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return n;
    };

    WorkSource work;
    work.init(P);
    LocalCounters lc = {0, 0, 0, 0, 0};
    uint32_t n_samples_done = 0;
    uint32_t dg_pass[kNumStatus] = {0, 0, 0, 0, 0, 0}, dg_lanes[kNumStatus] = {0, 0, 0, 0, 0, 0};
    uint32_t dg_steps = 0, dg_lane_steps = 0, dg_refills = 0, dg_census = 0;
    unsigned long long dg_t_trav = 0, dg_t_shade = 0, dg_t0 = 0, dg_tk = 0;
    uint32_t dg_leaf_rounds = 0, dg_leaf_lanes = 0, dg_walk_rounds = 0, dg_walk_lanes = 0;
    if (STATS) dg_t0 = __builtin_amdgcn_s_memtime();
    bool more_work = true;  // wave-uniform: the global work counter has not run out yet (as seen by this wave)
    const size_t npix = size_t(P.n_local_tiles) * 64u;
    const float eps = P.min_dist;

    // ---- per-lane traversal state; lives in registers across shading passes ----
    bool t_active = false;      // this lane is in the middle of a traversal
    bool t_has_result = false;  // this lane finished a traversal that is not finalised yet
    uint32_t t_slot = 0;
    V3 t_o = mk(0.0f, 0.0f, 0.0f), t_d = t_o;
    RayCull t_rc = {t_o, t_o, 0u, 16u, 32u, 0.0f, 0.0f};
    float t_best = 0.0f;
    uint32_t t_best_idx = 0, t_mesh = 0;
    const BvhNode4* t_nodes = nullptr;
    const BvhTri* t_tris = nullptr;
    uint32_t t_sp = 0;
    int32_t t_cur = 0;          // node to visit next: >= 0 inner, < 0 leaf, kNoChild = none (stack ran empty)
    int32_t t_pend = kNoChild;  // a leaf reached earlier whose triangles have not been tested yet
    uint32_t* const gstack = P.gstack + (size_t(blockIdx.x) * WAVES + wave) * kStackMax * 64u + lane;
    const uint32_t n_lds_stack = P.stack_entries;
    auto push = [&](int32_t v) {
        if (t_sp < n_lds_stack)
            stack[t_sp * 64u] = uint32_t(v);
        else
            gstack[(t_sp - n_lds_stack) * 64u] = uint32_t(v);
        ++t_sp;
    };
    auto pop = [&]() -> int32_t {
        --t_sp;
        return int32_t(t_sp < n_lds_stack ? stack[t_sp * 64u] : gstack[(t_sp - n_lds_stack) * 64u]);
    };

    uint32_t idle_spins = 0;
    for (;;) {
        if (fatal) break;
        // ---- finalise finished traversals in a batch (mesh.rs:245-266, scene.rs:33-41) ----
        if (__any(t_has_result)) {
            uint32_t next_k = 0xFFu;
            if (t_has_result) {
                const uint32_t slot = t_slot;
                float closest = __uint_as_float(WPOOL(F_DIST, slot));
                const uint32_t meta = WPOOL(F_META, slot);
                int32_t obj = int32_t((meta >> 14) & 255u) - 1;
                if (t_best > eps && t_best < 100000.0f) {  // triangle.rs:405
                    const V3 p = t_o + t_best * t_d;
                    const float dist = length(t_o - p);
                    if (dist > P.min_dist && dist < P.max_dist) {
                        if (STATS) ++lc.mesh_hits;
                        if (dist < closest) {
                            closest = dist;
                            obj = int32_t(P.n_spheres + t_mesh);
                            WPOOL(F_DIST, slot) = __float_as_uint(dist);
                            WPOOL(F_T, slot) = __float_as_uint(t_best);
                            WPOOL(F_TRI, slot) = t_best_idx;
                        }
                    }
                }
                const uint32_t m2 = next_gated_mesh<STATS>(sc, P.n_meshes, t_mesh + 1u, t_o, t_d, lc);
                const uint32_t depth = meta & 127u;
                WPOOL(F_META, slot) = pack_meta(depth, (meta >> 7) & 127u, obj, m2 < P.n_meshes ? m2 : 0u);
                next_k = m2 < P.n_meshes ? ST_TRAV : classify(sc, obj, depth);
                t_has_result = false;
            }
            q_push(next_k, t_slot);
        }
        // ---- queue depths (scheduling hints) ----
        uint32_t cnt[kNumStatus];
#pragma unroll
        for (uint32_t k = 0; k < kQ; ++k) cnt[k] = q_depth(k);
        if (STATS) ++dg_census;
        uint32_t n_active = uint32_t(__popcll(__ballot(t_active)));

        // ---- idle lanes take parked rays (in batches: only when enough lanes are idle) ----
        if (cnt[ST_TRAV] != 0 && (n_active < P.y_low_water || n_active + cnt[ST_TRAV] <= 64u)) {
            if (STATS) ++dg_refills;
            const uint64_t idle = __ballot(!t_active);
            // idle lanes are not contiguous: pop into lanes [0, n), then idle lane number r takes entry r
            uint32_t got_slot = 0;
            const uint32_t n_got = q_pop(ST_TRAV, uint32_t(__popcll(idle)), 0u, got_slot);
            const uint32_t rk = lane_rank(idle);
            const uint32_t handed = uint32_t(__shfl(int(got_slot), int(rk & 63u)));  // executed by every lane
            if (!t_active && rk < n_got) {
                const uint32_t slot = handed;
                t_slot = slot;
                t_o = mk(__uint_as_float(WPOOL(F_OX, slot)), __uint_as_float(WPOOL(F_OY, slot)),
                         __uint_as_float(WPOOL(F_OZ, slot)));
                t_d = mk(__uint_as_float(WPOOL(F_DX, slot)), __uint_as_float(WPOOL(F_DY, slot)),
                         __uint_as_float(WPOOL(F_DZ, slot)));
                t_mesh = (WPOOL(F_META, slot) >> 22) & 255u;
                const uint32_t* md = sc.mesh + t_mesh * kMeshDw;
                t_nodes = lds_ptr<BvhNode4>(md + MD_NODES);
                t_tris = lds_ptr<BvhTri>(md + MD_TRIS);
                t_rc = make_cull(t_o, t_d, reinterpret_cast<const float*>(md) + MD_CENTER, __uint_as_float(md[MD_RADIUS]),
                                 P.eps_frac);
                t_best = 1000000.0f;  // triangle.rs:398
                t_best_idx = 0;
                t_sp = 0;
                t_cur = 0;
                t_pend = kNoChild;
                t_active = true;
                if (STATS) ++dg_lanes[ST_TRAV];
            }
            cnt[ST_TRAV] -= n_got < cnt[ST_TRAV] ? n_got : cnt[ST_TRAV];
            n_active += n_got;
        }

        // ---- pick the shading kind with the deepest queue ----
        const uint32_t n_gen_slots = more_work ? cnt[ST_EMPTY] : 0u;
        uint32_t kind = ST_TERM, best = cnt[ST_TERM] + n_gen_slots;
        if (cnt[ST_LAMB] > best) kind = ST_LAMB, best = cnt[ST_LAMB];
        if (cnt[ST_METAL] > best) kind = ST_METAL, best = cnt[ST_METAL];
        if (cnt[ST_DIEL] > best) kind = ST_DIEL, best = cnt[ST_DIEL];
        if (best == 0 && n_active == 0) {
            // Nothing for this wave. Done when every slot is back in the EMPTY queue and no work is left;
            // otherwise other waves still own paths that may come this way: nap and look again.
            if (!more_work && cnt[ST_EMPTY] >= uint32_t(POOLN)) break;
            if (++idle_spins > (1u << 24)) {
                fatal = true;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
            continue;
        }
        idle_spins = 0;

        // Traverse while the lanes are well filled; shade when they are not (that is what parks new rays)
        // or when a full wave of shading work is waiting. Shallow queues are left to fill up while the
        // lanes have traversal work.
        const bool traverse =
            n_active != 0 && (best == 0 || (n_active >= P.y_low_water && best < 64u) || (best < P.shade_min && n_active >= 16u));
        if (traverse) {
            if (STATS) {
                ++dg_pass[ST_TRAV];
                if (best == 0) ++dg_pass[ST_EMPTY];
                dg_lanes[ST_EMPTY] += n_active;
                dg_tk = __builtin_amdgcn_s_memtime();
            }
            const uint32_t keep = (best != 0 || cnt[ST_TRAV] != 0) ? (n_active < P.y_low_water ? n_active : P.y_low_water)
                                                                   : n_active;
            uint32_t burst = 0;
            do {
                if (STATS) {
                    ++dg_steps;
                    dg_lane_steps += uint32_t(__popcll(__ballot(t_active)));
                }
                // Leaves are deferred: a lane that reaches a leaf remembers it (one pending leaf per lane) and
                // keeps walking; triangles are tested in rounds, when enough lanes hold a leaf or no lane can
                // walk on. Testing later only delays the shrinking of t_best, it cannot change the result.
                if (t_active && t_cur < 0 && t_cur != kNoChild && t_pend == kNoChild) {
                    t_pend = t_cur;
                    t_cur = t_sp != 0 ? pop() : kNoChild;
                }
                const bool can_walk = t_active && t_cur >= 0;
                // A lane is stalled when it holds a pending leaf and has reached another one (or the end of
                // its walk): it idles through every node round until the next leaf round. A leaf round is
                // run when enough lanes are stalled, when most lanes hold a leaf anyway, or when nobody can walk.
                const uint32_t n_pend = uint32_t(__popcll(__ballot(t_active && t_pend != kNoChild)));
                const uint32_t n_stalled = uint32_t(__popcll(__ballot(t_active && !can_walk)));
                if (STATS && __any(can_walk)) {
                    ++dg_walk_rounds;
                    dg_walk_lanes += uint32_t(__popcll(__ballot(can_walk)));
                }
                if (n_pend != 0 && (n_stalled >= P.leaf_round || n_pend >= 48u || !__any(can_walk))) {
                    if (STATS) {
                        ++dg_leaf_rounds;
                        dg_leaf_lanes += n_pend;
                    }
                    if (t_active && t_pend != kNoChild) {
                        leaf_test<STATS>(t_tris, t_pend, t_o, t_d, eps, P.eps_frac, t_best, t_best_idx, lc);
                        t_pend = kNoChild;
                    }
                }
                if (can_walk) {
                    uint32_t k[4];
                    f32x4 links;
                    node4_visit(t_nodes + t_cur, t_rc, eps, t_best, k, links);
                    if (STATS) ++lc.nodes;
                    if (k[0] != kMissKey) {  // farthest first, so that the nearest is popped first
                        if (k[3] != kMissKey) push(link_of(links, k[3]));
                        if (k[2] != kMissKey) push(link_of(links, k[2]));
                        if (k[1] != kMissKey) push(link_of(links, k[1]));
                        t_cur = link_of(links, k[0]);
                    } else {
                        t_cur = t_sp != 0 ? pop() : kNoChild;
                    }
                }
                if (t_active && t_cur == kNoChild && t_pend == kNoChild) {
                    t_active = false;
                    t_has_result = true;
                }
            } while (uint32_t(__popcll(__ballot(t_active))) >= keep && ++burst < 64u);
            if (STATS) dg_t_trav += __builtin_amdgcn_s_memtime() - dg_tk;
            continue;
        }

        // ============================ shading pass of one kind ===============================
        if (STATS) {
            ++dg_pass[kind];
            dg_tk = __builtin_amdgcn_s_memtime();
        }
        uint32_t slot = 0;
        const uint32_t n_main = q_pop(kind, 64u, 0u, slot);
        uint32_t n_gen = 0;
        if (kind == ST_TERM && n_gen_slots != 0 && n_main < 64u) n_gen = q_pop(ST_EMPTY, 64u - n_main, n_main, slot);
        const bool is_main = lane < n_main;
        const bool is_gen = !is_main && lane < n_main + n_gen;
        if (STATS) dg_lanes[kind] += n_main + n_gen;
        uint32_t next_k = 0xFFu;  // queue this lane's slot goes to after the pass

        V3 o = mk(0.0f, 0.0f, 0.0f), d = o;
        Rng rng = {0u, 0u};
        uint32_t item = 0, depth = 0, nrec = 0, word = 0;
        bool have_ray = false;

        if (kind == ST_TERM) {
            bool need_new = is_gen;
            if (is_main) {
                const uint32_t meta = WPOOL(F_META, slot);
                nrec = (meta >> 7) & 127u;
                const int32_t obj = int32_t((meta >> 14) & 255u) - 1;
                V3 color = mk(0.0f, 0.0f, 0.0f);  // hit with depth 0 or a failed scatter: lib.rs:63-66
                if (obj < 0) {                    // lib.rs:68-71, direction as is (not re-normalised)
                    const float dy = __uint_as_float(WPOOL(F_DY, slot));
                    const float t = 0.5f * (dy + 1.0f);
                    color = t * mk(1.0f, 1.0f, 1.0f) + (1.0f - t) * mk(P.bg);
                }
                if (nrec != 0) {  // lib.rs:62: attenuation * colorize(...), innermost bounce first
                    word = WPOOL(F_WORD, slot);
                    for (uint32_t k = nrec; k-- > 0;) {
                        const uint32_t w = (k >> 2) == (nrec >> 2) ? word : gseq[size_t(slot) * kSeqWords + (k >> 2)];
                        const uint32_t ob = (w >> (8u * (k & 3u))) & 0xFFu;
                        color = mk(reinterpret_cast<const float*>(sc.mat + ob * kMatDw)) * color;
                    }
                }
                item = WPOOL(F_ITEM, slot);  // index of this path's sample in the sample buffer
                float* out = P.sample_buf + size_t(item) * 3u;
                out[0] = color.x;
                out[1] = color.y;
                out[2] = color.z;
                if (STATS) ++n_samples_done;
                need_new = true;
            }
            // ---- new paths (cam.rs:64-82); work items come from the sharded global counters ----
            const uint64_t want = __ballot(need_new && more_work);
            if (want) {
                const uint32_t n_want = uint32_t(__popcll(want));
                const uint32_t avail = work.res_end - work.res_next;
                uint32_t new_lo = 0, new_hi = 0;
                if (avail < n_want) more_work = work.next_chunk(P, lane, new_lo, new_hi);
                if (need_new) {
                    const uint32_t rk = lane_rank(want);
                    const uint32_t it = rk < avail ? work.res_next + rk : new_lo + (rk - avail);
                    if (rk < avail || it < new_hi) {
                        const uint32_t pp = it & 63u;
                        const uint32_t ts = it >> 6;
                        const uint32_t tpos = div_magic(ts, P.batch, P.batch_magic);
                        const uint32_t s = ts - tpos * P.batch;
                        const uint32_t tile_local = P.tiles_reversed ? P.n_local_tiles - 1u - tpos : tpos;  // row-major, either way
                        item = s * uint32_t(npix) + tile_local * 64u + pp;       // < 2^32: the host sizes batches so
                        const uint32_t tile = tile_local * P.tile_world + P.tile_rank;
                        const uint32_t ty = div_magic(tile, P.tiles_x, P.tiles_x_magic), tx = tile - ty * P.tiles_x;
                        const uint32_t row = ty * RBRT_TILE + (pp >> 3), col = tx * RBRT_TILE + (pp & 7u);
                        if (row < P.cam.img_height_pix && col < P.cam.img_width_pix) {
                            rng.init(P.seed_key, row * P.cam.img_width_pix + col, P.sample_base + s);
                            const float col_off = float(col) - float(P.cam.img_width_pix / 2);
                            const float row_off = float(row) - float(P.cam.img_height_pix / 2);
                            const float u0 = rng.next_f32();
                            const float col_mm = ((col_off + u0) - 0.5f) * P.cam.mm_per_pix_hor;
                            const float u1 = rng.next_f32();
                            const float row_mm = ((row_off + u1) - 0.5f) * P.cam.mm_per_pix_vert;
                            const V3 pos = mk(P.cam.position);
                            const V3 target = (mk(P.cam.img_center_point) + (0.001f * col_mm) * mk(P.cam.right)) -
                                              (0.001f * row_mm) * mk(P.cam.up);
                            o = pos;
                            d = normalize(target - pos);
                            depth = P.max_depth;
                            nrec = 0;
                            word = 0;
                            have_ray = true;
                        }
                    }
                }
                if (avail < n_want) {
                    work.res_next = new_lo + (n_want - avail) < new_hi ? new_lo + (n_want - avail) : new_hi;
                    work.res_end = new_hi;
                } else {
                    work.res_next += n_want;
                }
                // reserve the next chunk now; its result is not needed before a later TERM pass
                if (more_work && work.res_end - work.res_next < 64u) work.prefetch(P, lane);
            }
        } else if (is_main) {
            // ---- RayScattering::scatter for one material kind (wave-uniform branch) ----
            o = mk(__uint_as_float(WPOOL(F_OX, slot)), __uint_as_float(WPOOL(F_OY, slot)),
                   __uint_as_float(WPOOL(F_OZ, slot)));
            d = mk(__uint_as_float(WPOOL(F_DX, slot)), __uint_as_float(WPOOL(F_DY, slot)),
                   __uint_as_float(WPOOL(F_DZ, slot)));
            rng.s0 = WPOOL(F_S0, slot);
            rng.s1 = WPOOL(F_S1, slot);
            item = WPOOL(F_ITEM, slot);
            word = WPOOL(F_WORD, slot);
            const uint32_t meta = WPOOL(F_META, slot);
            depth = meta & 127u;
            nrec = (meta >> 7) & 127u;
            const int32_t obj = int32_t((meta >> 14) & 255u) - 1;
            const float ht = __uint_as_float(WPOOL(F_T, slot));
            const V3 p = o + ht * d;  // same expression as inside the intersection routines
            V3 n;
            if (uint32_t(obj) < P.n_spheres) {
                n = p - mk(sc.sph + uint32_t(obj) * kSphDw);  // sphere.rs:56, unnormalised
            } else {
                const Normal4 nn =
                    lds_ptr<Normal4>(sc.mesh + (uint32_t(obj) - P.n_spheres) * kMeshDw + MD_NORMALS)[WPOOL(F_TRI, slot)];
                n = mk(nn.x, nn.y, nn.z);  // mesh.rs:253-257
            }
            DevMaterial m;
            {
                const uint32_t* mp = sc.mat + uint32_t(obj) * kMatDw;
                m.albedo[0] = __uint_as_float(mp[0]), m.albedo[1] = __uint_as_float(mp[1]);
                m.albedo[2] = __uint_as_float(mp[2]), m.param = __uint_as_float(mp[3]);
                m.kind = int32_t(mp[4]);
            }
            V3 nd;
            bool ok;
            if (kind == ST_LAMB) {  // lambertian.rs:11-24
                const V3 target = (p + normalize(n)) + random_point_in_unit_sphere(rng);
                nd = normalize(target - p);
                ok = true;
            } else if (kind == ST_METAL) {  // metal.rs:12-25
                const V3 target = reflect(d, n);
                nd = normalize(target + m.param * random_point_in_unit_sphere(rng));
                ok = dot(nd, n) > 0.0f;
            } else {  // dielectric.rs:11-59
                ok = scatter(m, d, p, n, rng, nd);
            }
            if (ok) {
                if (kind != ST_DIEL) {  // attenuation (1,1,1) is an exact identity, not recorded
                    word |= uint32_t(obj) << (8u * (nrec & 3u));
                    if ((nrec & 3u) == 3u) {
                        gseq[size_t(slot) * kSeqWords + (nrec >> 2)] = word;
                        word = 0;
                    }
                    ++nrec;
                }
                o = p;
                d = nd;
                depth -= 1;
                have_ray = true;
            } else {
                // metal.rs:25 returned false: the path is black (lib.rs:63-66). Park it for a TERM
                // pass with depth 0 so that the fold/store/regenerate code lives in one place.
                WPOOL(F_META, slot) = pack_meta(0u, nrec, obj, 0u);
                WPOOL(F_WORD, slot) = word;
                next_k = ST_TERM;
            }
        }

        // ---- closest sphere + mesh gate for the new ray (scene.rs:19-43 up to the meshes) ----
        if (have_ray) {
            if (STATS) ++lc.rays;
            float closest = 3.40282347e+38f, ht = 0.0f;
            int32_t obj = -1;
            for (uint32_t i = 0; i < P.n_spheres; ++i) {
                const float* sp = sc.sph + i * kSphDw;
                float t, dist;
                if (sphere_hit(mk(sp), sp[3], o, d, P.min_dist, P.max_dist, t, dist, P.counters)) {
                    if (dist < closest) {
                        closest = dist;
                        ht = t;
                        obj = int32_t(i);
                    }
                }
            }
            const uint32_t m = next_gated_mesh<STATS>(sc, P.n_meshes, 0, o, d, lc);
            WPOOL(F_OX, slot) = __float_as_uint(o.x);
            WPOOL(F_OY, slot) = __float_as_uint(o.y);
            WPOOL(F_OZ, slot) = __float_as_uint(o.z);
            WPOOL(F_DX, slot) = __float_as_uint(d.x);
            WPOOL(F_DY, slot) = __float_as_uint(d.y);
            WPOOL(F_DZ, slot) = __float_as_uint(d.z);
            WPOOL(F_S0, slot) = rng.s0;
            WPOOL(F_S1, slot) = rng.s1;
            WPOOL(F_ITEM, slot) = item;
            WPOOL(F_WORD, slot) = word;
            WPOOL(F_DIST, slot) = __float_as_uint(closest);
            WPOOL(F_T, slot) = __float_as_uint(ht);
            WPOOL(F_TRI, slot) = 0u;
            WPOOL(F_META, slot) = pack_meta(depth, nrec, obj, m < P.n_meshes ? m : 0u);
            next_k = m < P.n_meshes ? ST_TRAV : classify(sc, obj, depth);
        } else if (kind == ST_TERM && (is_main || is_gen)) {
            next_k = ST_EMPTY;  // no work item left (or a pixel outside a ragged image edge)
        }
        q_push(next_k, slot);
        if (STATS) dg_t_shade += __builtin_amdgcn_s_memtime() - dg_tk;
    }
#undef WPOOL
    if (fatal && lane == 0) atomicAdd(&P.counters->diag[23], 1ull);
    if (STATS) {
        atomicAdd(&P.counters->rays, (unsigned long long)lc.rays);
        atomicAdd(&P.counters->mesh_gate_pass, (unsigned long long)lc.gate);
        atomicAdd(&P.counters->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&P.counters->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&P.counters->mesh_hits, (unsigned long long)lc.mesh_hits);
        atomicAdd(&P.counters->samples, (unsigned long long)n_samples_done);
        if (lane == 0) {
            for (uint32_t k = 0; k < kNumStatus; ++k) {
                atomicAdd(&P.counters->diag[k], (unsigned long long)dg_pass[k]);
                atomicAdd(&P.counters->diag[6 + k], (unsigned long long)dg_lanes[k]);
            }
            atomicAdd(&P.counters->diag[12], (unsigned long long)dg_steps);
            atomicAdd(&P.counters->diag[13], (unsigned long long)dg_lane_steps);
            atomicAdd(&P.counters->diag[14], (unsigned long long)dg_refills);
            atomicAdd(&P.counters->diag[15], (unsigned long long)dg_census);
            atomicAdd(&P.counters->diag[16], dg_t_trav);
            atomicAdd(&P.counters->diag[17], dg_t_shade);
            atomicAdd(&P.counters->diag[18], (unsigned long long)(__builtin_amdgcn_s_memtime() - dg_t0));
            atomicAdd(&P.counters->diag[19], (unsigned long long)dg_leaf_rounds);
            atomicAdd(&P.counters->diag[20], (unsigned long long)dg_leaf_lanes);
            atomicAdd(&P.counters->diag[21], (unsigned long long)dg_walk_rounds);
            atomicAdd(&P.counters->diag[22], (unsigned long long)dg_walk_lanes);
        }
    }
}
