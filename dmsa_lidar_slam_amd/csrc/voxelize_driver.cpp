// voxelize_driver.cpp — currentGauss.reset() + createGaussianSets at both resolutions + Gaussian fit + updateRebalancingWeights
// (DmsaOptimizer.h:78-96) as one launch sequence over three streams; kernels in dmsa_kernels.hip / radix_sort.hip / serial_kernels.hip.
#include "dmsa_ctx.h"

// ---- Gaussians (DmsaOptimizer.h:78-96) ---------------------------------------------------------------------
// `overlap` (optional) runs on the host after every voxelisation kernel has been enqueued and before the counts are read
// back: host work placed there hides behind the GPU.
int build_gaussians(dmsa_ctx* ctx, const dmsa_settings& s, const std::function<int()>& overlap, bool allow_speculation, bool allow_compression, bool allow_small) {
    const int64_t n = ctx->n;
    ctx->gaussians_valid = false;
    ctx->order_valid = false;
    ctx->M = 0, ctx->M1 = 0, ctx->Mm = 0;
    const bool lvl_on[2] = {s.grid_size_1_factor > std::numeric_limits<float>::min(), s.grid_size_2_factor > std::numeric_limits<float>::min()};
    // createGaussianSets(set, factor * minGridSize, ...): float product, widened to double by the octree constructor
    ctx->level_res[0] = (double)(s.grid_size_1_factor * ctx->min_grid_size);
    ctx->level_res[1] = (double)(s.grid_size_2_factor * ctx->min_grid_size);
    if (!lvl_on[0]) ctx->level_res[0] = ctx->level_res[1];
    if (!lvl_on[1]) ctx->level_res[1] = ctx->level_res[0];
    const bool compress = ctx->compress_keys && allow_compression;
    // The reference's everyday size (5 scans x <= 3000 points + static points): ONE launch from the lattice to the member lists, a workgroup per
    // resolution (small_voxel.hip).  The kernel takes the tree depths and code widths from the lattice table on the device: nothing to
    // speculate on, no second stream, no joins.  Codes wider than 32 bits make it give up (counts.pad) and the general path runs.
    const bool small = allow_small && ctx->dbg.small_voxel != 0 && n <= small_voxel_max_points() && lvl_on[0] && lvl_on[1] &&
                       !(s.gauss_split != 0 && ctx->model == MODEL_KEYFRAMES) && ctx->dbg.voxel_coherence == 0;
    ctx->voxel_calls += 1;
    if (allow_speculation && ctx->dbg.speculation_fault > 0 && ctx->voxel_calls == ctx->dbg.speculation_fault)  // test hook: a guess one level too shallow
        for (int l = 0; l < 2; ++l) {
            if (ctx->depth_guess[l] > 1) ctx->depth_guess[l] -= 1;
            if (ctx->bits_guess[l] > 3) ctx->bits_guess[l] -= 3;
        }
    const bool speculate = small || (allow_speculation && ctx->depth_guess[0] >= 0 && ctx->depth_guess[1] >= 0 && ctx->depth_guess[0] < 20 && ctx->depth_guess[1] < 20 &&
                                     (!compress || (ctx->bits_guess[0] >= 0 && ctx->bits_guess[1] >= 0)));
    // the key kernels count the digits of the sort that follows (own sort, 32-bit codes): no clearing kernel, no histogram pass
    const bool prehist = ctx->prehist;
    const bool headers_zeroed = true;
    {
        ScopedTimer tm(ctx, T_VOXEL);
        const int nb = (int)((n + kAabbBlock - 1) / kAabbBlock);
        if (!ctx->aabb_fresh)  // the device loop's fused transform already left the block bounds and cleared the counters
            launch_block_aabb(ctx->d_global.as<float4>(), n, ctx->d_aabb.as<float>(), ctx->d_counts.p, sizeof(GaussCounts) + sizeof(SerialCounts),
                              ctx->stream);
        ctx->aabb_fresh = false;
        // the lattice kernel clears the headers of the own radix sorts on the side (one dispatch less per sort)
        // device-side stream dependencies (dev_sync.h) instead of events where a kernel of this sequence can carry the signal
        const bool dev_sync_lattice = !small && ctx->dbg.device_sync != 0 && ctx->dual_stream && lvl_on[0] && lvl_on[1];
        launch_lattice(ctx->d_global.as<float4>(), n, ctx->d_aabb.as<float>(), nb, ctx->level_res[0], ctx->level_res[1], compress,
                       ctx->d_lattice.as<LatticeTable>(), headers_zeroed ? ctx->d_sort_tmp[0].p : nullptr, headers_zeroed ? ctx->d_sort_tmp[1].p : nullptr, ctx->stream,
                       dev_sync_lattice ? ctx->sync_counter(SYNC_LATTICE) : nullptr, ctx->dbg.lattice_hint != 0 && ctx->lattice_hint_valid);
        ctx->lattice_hint_valid = true;  // (a table of another problem is harmless: it fails the verification and the replay runs)
        if (dev_sync_lattice) ctx->sync_sig[SYNC_LATTICE] += 2;  // one per resolution
        if (!speculate) {  // sync #1: tree depths select the radix-sort bit range (speculation reads them with the counts instead)
            HIPCHK(hipMemcpyAsync(ctx->h_lattice, ctx->d_lattice.p, 2 * sizeof(LatticeTable), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(sync_spin(ctx->stream));
        }
    }
    // The sort only needs an UPPER bound of the tree depth.  From the second iteration on the previous depths are used
    // without waiting for the lattice kernel; the true depths arrive with the counts (sync #2) and a too-small guess (the
    // bounding box doubled between two iterations) re-runs the voxelisation synchronously.
    int sort_depth[2], sort_bits[2];
    for (int l = 0; l < 2; ++l) {
        sort_depth[l] = speculate ? ctx->depth_guess[l] : ctx->h_lattice[l].final_depth;
        // width of the leaf codes: all 3*depth bits, or (compressed) only the bits that vary over the points
        sort_bits[l] = !compress ? 3 * sort_depth[l] : (speculate ? ctx->bits_guess[l] : ctx->h_lattice[l].total_bits);
        if (!speculate && lvl_on[l] && ctx->h_lattice[l].status != 0) return ctx->h_lattice[l].status;
    }
    GaussCounts* counts = ctx->d_counts.as<GaussCounts>();
    const bool split = s.gauss_split != 0 && ctx->model == MODEL_KEYFRAMES;
    if (split && !ctx->d_split_stats.p) {
        HIPCHK(ctx->d_split_stats.ensure(64 * 16));  // 64 stripes of (blocks looked at, blocks skipped)
        HIPCHK(hipMemsetAsync(ctx->d_split_stats.p, 0, 64 * 16, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // (both level streams add to it)
    }
    if (split)
        for (int l = 0; l < 2; ++l) {
            HIPCHK(ctx->d_pos_slot_rank[l].ensure((size_t)n * 4));
            HIPCHK(ctx->d_nsorted[l].ensure((size_t)n * 16));
            HIPCHK(ctx->d_pair_d[l].ensure(split_scratch_bytes(n)));
        }
    // The two resolutions are independent until their member lists are appended (level 1 starts at level 0's totals):
    // level 0 runs on `stream`, level 1 on `stream2`; their launches are enqueued stage by stage so that both streams fill.
    const bool two = ctx->dual_stream && lvl_on[0] && lvl_on[1];
    hipStream_t st[2] = {ctx->stream, two ? ctx->stream2 : ctx->stream};
    bool k32v[2] = {false, false};
    // Both resolutions are keyed into one array of 2n (code, point) pairs -- level 1 carries a tag bit above the widest code --
    // and sorted by ONE radix sort: half the launches, twice the parallelism per pass, and the sorted halves are the two levels.
    if (ctx->dbg.trace_time >= 3) std::fprintf(stderr, "[voxelize] key bits (without the marker of non-finite points): level 0 %d, level 1 %d\n", sort_bits[0], sort_bits[1]);
    const int tag_bit = std::max(sort_bits[0], sort_bits[1]) + 1;  // bit `sort_bits` is the marker of non-finite points
    const unsigned end_bit = (unsigned)(tag_bit + 1);
    const bool k32 = small || end_bit <= 32;  // (the small path writes 32-bit codes or gives up)
    {
        const size_t ksz = k32 ? 4 : 8;
        for (int l = 0; l < 2; ++l) {
            k32v[l] = ctx->key32[l] = k32;
            ctx->code_v[l] = ctx->d_code[0].as<char>() + (size_t)l * n * ksz, ctx->code_s_v[l] = ctx->d_code_s[0].as<char>() + (size_t)l * n * ksz;
            ctx->idx_v[l] = ctx->d_idx[0].as<uint32_t>() + (size_t)l * n, ctx->idx_s_v[l] = ctx->d_idx_s[0].as<uint32_t>() + (size_t)l * n;
        }
    }
    // Small clouds (keyframe sets: 3 x 10^5 points) are launch-bound: one sort of 2n pairs on one stream.  Large clouds (the window:
    // 1.5 x 10^6) keep the two levels on two streams with one sort each (a sort only looks at the bits below its end bit, so the
    // tag is inert there).
    const bool merged = ctx->merge_sort < 0 ? n <= (int64_t)(1 << 20) : ctx->merge_sort != 0;
    const bool prepared = prehist && k32;
    SortPlan plan[2];  // merged: one sort of 2n pairs in workspace 0, both key kernels count into its header
    for (int l = 0; l < 2 && !small; ++l)
        plan[l] = merged ? sort_pairs_u32_plan(ctx->d_sort_tmp[0].p, (size_t)(2 * n), end_bit)
                         : sort_pairs_u32_plan(ctx->d_sort_tmp[l].p, (size_t)n, (unsigned)(sort_bits[l] + 1));
    auto stage_keys = [&](int l, hipStream_t stream) {  // a disabled level is keyed with the other level's lattice (level_res is aliased) and ignored later
        SortPlan pl = plan[l];
        if (merged && l == 1) pl.state_words = 0;  // the look-back words of the common sort are cleared once
        launch_voxel_keys(ctx->d_global.as<float4>(), n, ctx->d_lattice.as<LatticeTable>() + l, ctx->level_res[l], ctx->code_v[l], k32, ctx->idx_v[l],
                          l == 0 ? 0ull : (1ull << tag_bit), prepared ? &pl : nullptr, stream);
        if (ctx->dbg.voxel_coherence != 0 && lvl_on[l]) {  // how many points changed leaf since the previous voxelisation of this context
            const size_t ksz = k32 ? 4 : 8;
            const bool compare = ctx->coh_valid[l] && ctx->d_code_prev[l].cap >= (size_t)n * ksz && ctx->coh_key32[l] == k32 && ctx->coh_n[l] == n;
            if (ctx->d_code_prev[l].ensure((size_t)n * ksz) != hipSuccess || ctx->d_coh_count.ensure(8) != hipSuccess) return;
            if (!ctx->coh_count_zeroed) (void)hipMemsetAsync(ctx->d_coh_count.p, 0, 8, stream), ctx->coh_count_zeroed = true;
            launch_count_code_changes(ctx->code_v[l], ctx->d_code_prev[l].p, k32, n, compare, ctx->d_coh_count.as<unsigned long long>(), stream);
            if (compare) ctx->coh_compared += n;
            ctx->coh_pending[l] = compare;
            ctx->coh_valid[l] = true, ctx->coh_key32[l] = k32, ctx->coh_n[l] = n;
        }
    };
    auto stage_sort_both = [&]() -> int {
        stage_keys(0, ctx->stream), stage_keys(1, ctx->stream);
        if (k32 && prepared)
            HIPCHK(sort_pairs_u32_onesweep(ctx->d_sort_tmp[0].p, ctx->d_sort_tmp[0].cap, ctx->d_code[0].as<uint32_t>(), ctx->d_code_s[0].as<uint32_t>(),
                                           ctx->d_idx[0].as<uint32_t>(), ctx->d_idx_s[0].as<uint32_t>(), (size_t)(2 * n), end_bit, ctx->stream, 2));
        else if (k32)
            HIPCHK(sort_pairs_u32_u32(ctx->d_sort_tmp[0].p, ctx->d_sort_tmp[0].cap, ctx->d_code[0].as<uint32_t>(), ctx->d_code_s[0].as<uint32_t>(),
                                      ctx->d_idx[0].as<uint32_t>(), ctx->d_idx_s[0].as<uint32_t>(), (size_t)(2 * n), end_bit, ctx->stream, headers_zeroed));
        else
            HIPCHK(sort_pairs_u64_u32(ctx->d_sort_tmp[0].p, ctx->d_sort_tmp[0].cap, ctx->d_code[0].as<uint64_t>(), ctx->d_code_s[0].as<uint64_t>(),
                                      ctx->d_idx[0].as<uint32_t>(), ctx->d_idx_s[0].as<uint32_t>(), (size_t)(2 * n), end_bit, ctx->stream));
        return DMSA_OK;
    };
    auto stage_sort = [&](int l) -> int {
        stage_keys(l, st[l]);
        const unsigned eb = (unsigned)(sort_bits[l] + 1);
        if (k32 && prepared)
            HIPCHK(sort_pairs_u32_onesweep(ctx->d_sort_tmp[l].p, ctx->d_sort_tmp[l].cap, (const uint32_t*)ctx->code_v[l], (uint32_t*)ctx->code_s_v[l], ctx->idx_v[l],
                                           ctx->idx_s_v[l], (size_t)n, eb, st[l], 2));
        else if (k32)
            HIPCHK(sort_pairs_u32_u32(ctx->d_sort_tmp[l].p, ctx->d_sort_tmp[l].cap, (const uint32_t*)ctx->code_v[l], (uint32_t*)ctx->code_s_v[l], ctx->idx_v[l],
                                      ctx->idx_s_v[l], (size_t)n, eb, st[l], headers_zeroed));
        else
            HIPCHK(sort_pairs_u64_u32(ctx->d_sort_tmp[l].p, ctx->d_sort_tmp[l].cap, (const uint64_t*)ctx->code_v[l], (uint64_t*)ctx->code_s_v[l], ctx->idx_v[l],
                                      ctx->idx_s_v[l], (size_t)n, eb, st[l]));
        return DMSA_OK;
    };
    auto stage_leaves = [&](int l) -> int {
        const LatticeTable* tab = ctx->d_lattice.as<LatticeTable>() + l;
        const bool k32 = k32v[l];
        if (ctx->fused_segments) {
            // one single-pass kernel; its look-back state is never cleared (epoch-tagged words, running ticket counter)
            const size_t need = 8 * (size_t)(1 + leaf_segment_tiles(n));
            if (need > ctx->d_seg_state[l].cap) {
                HIPCHK(ctx->d_seg_state[l].ensure(need));
                HIPCHK(hipMemsetAsync(ctx->d_seg_state[l].p, 0, ctx->d_seg_state[l].cap, st[l]));
                ctx->seg_epoch[l] = 0, ctx->seg_ticket[l] = 0;
            }
            ctx->seg_epoch[l] += 1;
            if (ctx->seg_epoch[l] == 0) ctx->seg_epoch[l] = 1;
            launch_leaf_segments(ctx->code_s_v[l], k32, n, tab, ctx->d_leaf_incl[l].as<int32_t>(), ctx->d_leaf_start[l].as<int32_t>(), &counts->level[l],
                                 ctx->d_seg_state[l].as<unsigned long long>(), ctx->seg_epoch[l], ctx->seg_ticket[l], st[l]);
            ctx->seg_ticket[l] += (uint32_t)leaf_segment_tiles(n);
        } else {
            launch_head_flags(ctx->code_s_v[l], k32, n, tab, ctx->d_head[l].as<int32_t>(), st[l]);
            HIPCHK(inclusive_scan_i32(ctx->d_scan_tmp[l].p, ctx->d_scan_tmp[l].cap, ctx->d_head[l].as<int32_t>(), ctx->d_leaf_incl[l].as<int32_t>(), (size_t)n, st[l]));
            launch_leaf_starts(ctx->d_head[l].as<int32_t>(), ctx->d_leaf_incl[l].as<int32_t>(), ctx->code_s_v[l], k32, tab, n,
                               ctx->d_leaf_start[l].as<int32_t>(), &counts->level[l], st[l]);
        }
        // slot scans: one multi-workgroup single-pass kernel (debug switch fused_leaf_scan = 0: the single-workgroup k_leaf_scan)
        auto finalize = [&]() -> int {
            const size_t need = leaf_finalize_state_bytes(n);
            if (need > ctx->d_fin_state[l].cap) {
                HIPCHK(ctx->d_fin_state[l].ensure(need));
                HIPCHK(hipMemsetAsync(ctx->d_fin_state[l].p, 0, ctx->d_fin_state[l].cap, st[l]));
                ctx->fin_epoch[l] = 0, ctx->fin_ticket[l] = 0;
            }
            ctx->fin_epoch[l] += 1;
            if (ctx->fin_epoch[l] == 0) ctx->fin_epoch[l] = 1;
            launch_leaf_finalize(ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(), n, ctx->d_gauss_of_slot[l].as<int32_t>(),
                                 ctx->d_memb_of_slot[l].as<int32_t>(), ctx->d_pslot_of_slot[l].as<int32_t>(), &counts->level[l],
                                 ctx->d_fin_state[l].as<unsigned long long>(), ctx->fin_epoch[l], ctx->fin_ticket[l], st[l]);
            ctx->fin_ticket[l] += (uint32_t)leaf_finalize_tiles(n);
            return DMSA_OK;
        };
        launch_leaf_accept(ctx->d_leaf_start[l].as<int32_t>(), ctx->idx_s_v[l], ctx->d_ring.as<int32_t>(), &counts->level[l],
                           s.min_num_points_per_set, n, ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(), st[l]);
        if (split)
            launch_leaf_split(ctx->d_leaf_incl[l].as<int32_t>(), ctx->d_leaf_start[l].as<int32_t>(), ctx->idx_s_v[l], ctx->d_ring.as<int32_t>(),
                              ctx->d_nglobal.as<float4>(), &counts->level[l], s.min_num_points_per_set, n, ctx->d_nsorted[l].as<float4>(),
                              ctx->d_pair_d[l].as<unsigned long long>(), ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(),
                              ctx->d_pos_slot_rank[l].as<int32_t>(), st[l], ctx->dbg.skip_stats != 0 ? ctx->d_split_stats.as<unsigned long long>() : nullptr);
        if (ctx->dbg.fused_leaf_scan != 0)
            CHK(finalize());
        else
            launch_leaf_scan(ctx->d_slot_acc[l].as<int32_t>(), ctx->d_slot_cnt[l].as<int32_t>(), ctx->d_gauss_of_slot[l].as<int32_t>(), ctx->d_memb_of_slot[l].as<int32_t>(),
                             ctx->d_pslot_of_slot[l].as<int32_t>(), &counts->level[l], st[l]);
        return DMSA_OK;
    };
    auto stage_gather = [&](int l, hipStream_t gs) {
        const LatticeTable* tab = ctx->d_lattice.as<LatticeTable>() + l;
        launch_gather_members(ctx->d_leaf_incl[l].as<int32_t>(), ctx->d_leaf_start[l].as<int32_t>(), ctx->idx_s_v[l],
                              ctx->code_s_v[l], k32v[l], tab, ctx->d_slot_acc[l].as<int32_t>(), ctx->d_gauss_of_slot[l].as<int32_t>(),
                              ctx->d_memb_of_slot[l].as<int32_t>(), split ? ctx->d_pos_slot_rank[l].as<int32_t>() : nullptr, ctx->d_local.as<float4>(),
                              ctx->d_slot_cnt[l].as<int32_t>(), counts, l, n, ctx->d_memb_local.as<float4>(), ctx->d_memb_idx.as<int32_t>(),
                              ctx->d_memb_g.as<int32_t>(), ctx->d_seg_off.as<int32_t>(), ctx->d_pslot_of_slot[l].as<int32_t>(), ctx->d_pad_off.as<int32_t>(), gs);
    };
    if (small) {
        ScopedTimer tm(ctx, T_VOXEL);
        SmallVoxelArgs a{};
        a.global = ctx->d_global.as<float4>(), a.local = ctx->d_local.as<float4>(), a.ring = ctx->d_ring.as<int32_t>();
        a.n = (int)n, a.min_pts = s.min_num_points_per_set;
        a.tables = ctx->d_lattice.as<LatticeTable>();
        for (int l = 0; l < 2; ++l) {
            a.res[l] = ctx->level_res[l];
            a.code[l] = (uint32_t*)ctx->code_v[l], a.idx[l] = ctx->idx_v[l], a.code_s[l] = (uint32_t*)ctx->code_s_v[l], a.idx_s[l] = ctx->idx_s_v[l];
        }
        a.counts = counts;
        a.memb_local = ctx->d_memb_local.as<float4>(), a.memb_idx = ctx->d_memb_idx.as<int32_t>(), a.memb_g = ctx->d_memb_g.as<int32_t>();
        a.seg_off = ctx->d_seg_off.as<int32_t>();
        a.sync = ctx->sync_counter(SYNC_SMALL_L0);
        ctx->sync_sig[SYNC_SMALL_L0] += 1;
        a.sync_target = ctx->sync_sig[SYNC_SMALL_L0];
        a.timed_out = ctx->sync_timed_out();
        if (ctx->dbg.gap_stamps == 3) {  // phase stamps of the kernel (device wall clock, 100 MHz), printed after the counts have arrived
            HIPCHK(ctx->d_sv_stamps.ensure(2 * 16 * 8));
            a.stamps = ctx->d_sv_stamps.as<long long>();
        }
        launch_voxel_small(a, ctx->stream);
        ctx->small_voxel_launches += 1;
        HIPCHK(hipGetLastError());
    } else {
        ScopedTimer tm(ctx, T_VOXEL);
        const bool dev_sync = two && ctx->dbg.device_sync != 0;
        if (merged) CHK(stage_sort_both());
        if (dev_sync) {
            // level 1 on its own stream: it needs the lattice (k_lattice signals) and, merged, the common sort (a signal kernel behind it)
            if (merged) {
                launch_sync_signal(ctx->sync_counter(SYNC_LATTICE), ctx->stream);
                ctx->sync_sig[SYNC_LATTICE] += 1;
            }
            enqueue_wait(ctx, SYNC_LATTICE, ctx->stream2);
        } else if (two) {
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        }
        if (!merged)
            for (int l = 0; l < 2; ++l)
                if (lvl_on[l]) CHK(stage_sort(l));
        // (Enqueueing level 1 first -- its chain ends last -- was measured in round 5: 1089 instead of 1259 it/s.  Level 1's first kernel sits behind
        // a device-side wait; launched ahead of level 0's kernels the wait's queue is served first and level 0 starts late.
        // Swapping the streams as well -- level 1 on the main stream and enqueued first, level 0 behind the wait on the side stream -- was also
        // measured: 1243 against 1263 it/s on the headline window, the small windows and the keyframe set unchanged.)
        for (int l = 0; l < 2; ++l) {
            if (!lvl_on[l]) continue;
            CHK(stage_leaves(l));
        }        // Both gathers on the first stream (level 1 appends behind level 0's totals anyway): the level-1 chain ends with its leaf scan,
        // long before level 0's gather is through, so the wait below finds its event signalled -- a join at the END of a stream costs
        // ~20 us of cross-queue signalling in front of everything that follows.
        if (dev_sync) {
            launch_sync_signal(ctx->sync_counter(SYNC_LEVEL1), ctx->stream2);
            ctx->sync_sig[SYNC_LEVEL1] += 1;
        } else if (two) {
            HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        }
        if (lvl_on[0]) stage_gather(0, ctx->stream);
        if (dev_sync)
            enqueue_wait(ctx, SYNC_LEVEL1, ctx->stream);
        else if (two)
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        if (lvl_on[1]) stage_gather(1, ctx->stream);
    }
    // The read-back of the counts runs on the third stream: a device-to-host copy ends with a system-scope release that holds up the
    // stream it is on for ~20 us, and the fit behind it does not need to wait for that.
    hipStream_t rb = ctx->dual_stream ? ctx->stream3 : ctx->stream;
    bool rb_released = false;
    // Which Gaussians walk their members on one lane group (k_residuals_small, the fit's one-wave class) and which get a workgroup
    // (chain tiers, the fit's four-wave class)?  The lane-per-evaluation kernel spends the fewest instructions per member, but a wave of
    // it walks up to `threshold` members one after the other: with a few thousand Gaussians (the reference's everyday windows, small
    // keyframe sets) its few hundred waves leave the chip idle while the longest of them runs.  Measured (it/s at thresholds 8 = nobody /
    // 32 / 256, scripts/threshold_ab.sh): config-2 window, 1100 Gaussians 3450 / 3090 / 2580; 8 keyframes, 2465: 1965 / 1955 / 1906;
    // 12 keyframes, 3823: 1672 / 1782 / 1726; 16 keyframes, 4920: 1353 / 1534 / 1502; 24 keyframes, 6906: 910 / 1080 / 1213; the bench
    // window, 13 000: - / 1196 (64) / 1262.  Every tier computes the same bits, so the rule may follow the size of the previous
    // voxelisation (the first one of a context goes by the number of points).
    const int auto_threshold = ctx->M > 0 ? (ctx->M < 2000 ? 8 : ctx->M < 6000 ? 32 : 0) : (n < 60000 ? 8 : n < 200000 ? 32 : 0);
    const int small_threshold = ctx->dbg.small_threshold > 0 ? ctx->dbg.small_threshold : auto_threshold;
    {   // size classes of the correspondence kernels: needs only seg_off, so it runs before the read-back
        // k_size_classes is one workgroup on the main stream between the voxelisation and the fit: it also carries two stream dependencies
        // (dev_sync.h) -- it waits for the pose tables of the Jacobian batch (built on the side stream long ago) and releases the read-back
        DevSync sy;
        if (ctx->dbg.device_sync != 0) {
            sy.timed_out = ctx->sync_timed_out();
            if (ctx->tables_pending && ctx->tables_dev_sync) {
                sy.wait_counter = ctx->sync_counter(SYNC_TABLES), sy.wait_target = ctx->sync_sig[SYNC_TABLES];
                ctx->tables_pending = false;
            }
            if (rb != ctx->stream) sy.signal_counter = ctx->sync_counter(SYNC_CLASSES), ctx->sync_sig[SYNC_CLASSES] += 1, rb_released = true;
        }
        launch_size_classes(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_order.as<uint32_t>(),
                            reinterpret_cast<SerialCounts*>(ctx->d_counts.as<char>() + sizeof(GaussCounts)), ctx->stream, sy, small_threshold, ctx->dbg.long_log2);
    }
    if (ctx->stamp_voxel) launch_stamp(ctx->stamp_voxel, ctx->stream);
    if (rb_released) {
        enqueue_wait(ctx, SYNC_CLASSES, rb);
    } else if (rb != ctx->stream) {
        HIPCHK(hipEventRecord(ctx->ev_scan0, ctx->stream));
        HIPCHK(hipStreamWaitEvent(rb, ctx->ev_scan0, 0));
    }
    HIPCHK(hipMemcpyAsync(&ctx->h_rb->g, ctx->d_counts.p, sizeof(GaussCounts) + sizeof(SerialCounts), hipMemcpyDeviceToHost, rb));
    HIPCHK(hipMemcpyAsync(ctx->h_lattice, ctx->d_lattice.p, 2 * sizeof(LatticeTable), hipMemcpyDeviceToHost, rb));  // incl. out_of_range
    if (ctx->rb_extra_bytes)  // device loop: the previous iteration's stop decision travels with the counts
        HIPCHK(hipMemcpyAsync(ctx->rb_extra_dst, ctx->rb_extra_src, ctx->rb_extra_bytes, hipMemcpyDeviceToHost, rb));
    HIPCHK(hipEventRecord(ctx->ev_counts, rb));
    // The fit is enqueued BEHIND the read-back, with the previous iteration's class
    // counts (+ margin) as grids -- the kernels take the true ranges from device memory, surplus workgroups exit, and whatever the
    // guess missed is launched after sync #2.  The three classes run side by side on two streams (each is latency-bound on its own).
    const int32_t* d_sc = reinterpret_cast<const int32_t*>(ctx->d_counts.as<char>() + sizeof(GaussCounts));
    const float* fit_table = ctx->base_table ? ctx->base_table : ctx->d_tables.as<float>();
    int fit_launched[3] = {0, 0, 0}, finish_launched = 0;
    auto launch_fit = [&](const int first[3], const int tasks_in[3], int finish_gauss) -> int {
        ScopedTimer tm(ctx, T_FIT);
        const int tasks[3] = {(ctx->dbg.fit_classes & 1) ? tasks_in[0] : 0, (ctx->dbg.fit_classes & 2) ? tasks_in[1] : 0, (ctx->dbg.fit_classes & 4) ? tasks_in[2] : 0};
        {
            // one launch for the three size classes and the rebalancing weights: no fork to a second stream, no join
            launch_gauss_fit_all(ctx->d_memb_local.as<float4>(), ctx->d_seg_off.as<int32_t>(), fit_table, ctx->d_order.as<uint32_t>(), d_sc, first, tasks,
                                 ctx->d_fit_sums.as<float>(), counts, ctx->d_info12.as<float>(), true, ctx->rows - 1, ctx->d_gauss_rows.as<int2>(), ctx->dbg.eigen_l1_bytes,
                                 ctx->d_pow_codes.as<uint32_t>(), (int)std::min<int64_t>(ctx->pow_n, INT32_MAX), ctx->d_memb_q.as<float>(), (size_t)(2 * ctx->n + 16), ctx->stream);
        }
        launch_gauss_fit_finish(ctx->d_seg_off.as<int32_t>(), counts, ctx->d_fit_sums.as<float>(), finish_gauss, ctx->d_info12.as<float>(), ctx->stream);
        HIPCHK(hipGetLastError());
        return DMSA_OK;
    };
    if (ctx->fit_guess_valid) {
        const SerialCounts& pg = ctx->serial_counts;  // previous iteration
        const int first[3] = {0, 0, 0};
        auto grow = [](int v) { return v + v / 8 + 16; };
        fit_launched[0] = grow(pg.n_long), fit_launched[1] = grow(pg.n_chain - pg.n_long), fit_launched[2] = grow(pg.n_small);
        finish_launched = grow(pg.n_chain + pg.n_small);
        CHK(launch_fit(first, fit_launched, finish_launched));
        if (ctx->stamp_fit) launch_stamp(ctx->stamp_fit, ctx->stream);
    }
    g_tl.mark("voxel enq");
    if (overlap) CHK(overlap());
    g_tl.mark("jacobian batch host+enq");
    {  // sync #2: M sizes every later launch
        hipError_t e;
        while ((e = hipEventQuery(ctx->ev_counts)) == hipErrorNotReady) {
        }
        (void)hipGetLastError();  // see sync_spin
        HIPCHK(e);
    }
    g_tl.mark("sync#2 wait");
    const GaussCounts h = ctx->h_rb->g;
    if (small && ctx->dbg.gap_stamps == 3 && ctx->d_sv_stamps.p) {
        long long st[32];
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(st, ctx->d_sv_stamps.p, sizeof(st), hipMemcpyDeviceToHost));
        static int printed = 0;
        if (printed++ % 16 == 8) {
            for (int l = 0; l < 2; ++l) {
                std::fprintf(stderr, "[k_voxel_small level %d] us since its start:", l);
                for (int k = 1; k < 12; ++k) std::fprintf(stderr, " %.1f", (double)(st[l * 16 + k] - st[l * 16]) / 100.0);
                std::fprintf(stderr, "   (level start offset %.1f us)\n", (double)(st[l * 16] - st[0]) / 100.0);
            }
        }
    }
    if (small && (h.pad[0] != 0 || h.pad[1] != 0)) {  // codes wider than 32 bits (or a lattice error): the general path sorts them out
        if (ctx->h_lattice[0].status != 0) return ctx->h_lattice[0].status;
        if (ctx->h_lattice[1].status != 0) return ctx->h_lattice[1].status;
        ctx->small_voxel_fallbacks += 1;
        ctx->depth_guess[0] = ctx->depth_guess[1] = -1;
        return build_gaussians(ctx, s, nullptr, false, allow_compression, false);
    }
    for (int l = 0; l < 2; ++l) {
        if (lvl_on[l] && ctx->h_lattice[l].status != 0) return ctx->h_lattice[l].status;
        const int true_bits = compress ? ctx->h_lattice[l].total_bits : 3 * ctx->h_lattice[l].final_depth;
        if (lvl_on[l] && compress && ctx->h_lattice[l].out_of_range) {
            ctx->depth_guess[0] = ctx->depth_guess[1] = -1;
            ctx->speculation_retries += 1;
            return build_gaussians(ctx, s, nullptr, false, false, allow_small);  // a key left the predicted range: redo with full-width codes
        }
        if (!small && speculate && lvl_on[l] && (ctx->h_lattice[l].final_depth > sort_depth[l] || true_bits > sort_bits[l])) {
            ctx->depth_guess[0] = ctx->depth_guess[1] = -1;
            ctx->speculation_retries += 1;
            return build_gaussians(ctx, s, nullptr, false, allow_compression, allow_small);  // mis-speculated: redo (overlap work already ran)
        }
        if (lvl_on[l]) (ctx->h_lattice[l].pad3 ? ctx->lattice_hints_held : ctx->lattice_replays) += 1;
#ifdef DMSA_LATTICE_TIMING
        if (ctx->dbg.host_timeline != 0 && lvl_on[l]) {
            std::fprintf(stderr, "[k_lattice level %d hint %d] x10ns:", l, ctx->h_lattice[l].pad3);
            for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %.0f", ctx->h_lattice[l].mn[kMaxLatticeEvents - 8 + k][0]);
            std::fprintf(stderr, "\n");
        }
#endif
        ctx->depth_guess[l] = ctx->h_lattice[l].final_depth;
        ctx->bits_guess[l] = ctx->h_lattice[l].total_bits;
        if (ctx->dbg.voxel_coherence != 0 && lvl_on[l]) {  // a different lattice re-labels every leaf: such a voxelisation could not reuse the previous order
            const LatticeTable &a = ctx->h_lattice[l], &b = ctx->coh_lattice[l];
            const bool same = a.final_depth == b.final_depth && a.compressed == b.compressed && std::memcmp(a.final_mn, b.final_mn, sizeof(a.final_mn)) == 0 &&
                              std::memcmp(a.nbits, b.nbits, sizeof(a.nbits)) == 0 && std::memcmp(a.key_base, b.key_base, sizeof(a.key_base)) == 0;
            if (ctx->coh_pending[l] && !same) ctx->coh_lattice_changes += 1;
            ctx->coh_lattice[l] = a, ctx->coh_pending[l] = false;
        }
    }
    {
        ScopedTimer tm(ctx, T_FIT);
        const int M_all = h.level[0].num_gauss + h.level[1].num_gauss;
        // whatever the pre-sync launches did not cover (first iteration, or a class that grew by more than the margin)
        ctx->serial_counts = ctx->h_rb->sc;
        const SerialCounts& sc = ctx->serial_counts;
        const int want[3] = {sc.n_long, sc.n_chain - sc.n_long, sc.n_small};
        int rest[3], any = 0;
        for (int c = 0; c < 3; ++c) rest[c] = std::max(0, want[c] - fit_launched[c]), any += rest[c];
        if (M_all > 0 && (any > 0 || finish_launched < M_all)) CHK(launch_fit(fit_launched, rest, M_all));
        ctx->fit_guess_valid = M_all > 0;
        ctx->order_valid = true;
    }
    HIPCHK(hipGetLastError());
    ctx->M1 = h.level[0].num_gauss;
    ctx->M = h.level[0].num_gauss + h.level[1].num_gauss;
    ctx->Mm = (int64_t)h.level[0].num_memb + h.level[1].num_memb;
    ctx->gaussians_valid = true;
    return DMSA_OK;
}

