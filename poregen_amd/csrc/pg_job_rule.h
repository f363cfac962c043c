// pg_job_rule.h -- where a rank's statistics go relative to the count exchange of a multi-GPU step (pg_job.hip, job_exchange_and_collect).
// Host-only arithmetic, kept apart so that it can be tested without a device (poregen_amd/_pg_hosttest.so: pgt_job_stats_rule;
// tests/test_host_logic.py). PROVISIONAL: no run on more than one GPU exists in this pool ("Multi-GPU"); the rule can be
// overridden per process with PGMOVE_JOB_STATS_RULE=front|behind|auto.
//
// The two places:
//   FRONT  the statistics (k_read_stats, the largest HBM-bound kernel of a batch) are queued before the stream waits for the exchanged
//          table: they hide the collective and the slowest rank's counting chain. A rank that turns out to keep nothing (every k-mer
//          complete below it: the reference would not have read its lines, gmove.cpp:733-735) has then computed them for nothing.
//   BEHIND they are queued behind the wait and cancelled on the device when the table says the rank keeps nothing (k_base_open /
//          k_stats_cancel): nothing is wasted, nothing is hidden.
// BEHIND pays in the one batch that completes the job, FRONT in every other batch. So: BEHIND only where completion below the rank is
// plausible in THIS batch -- the ops of the lower ranks, spread evenly over the k-mers, could fill what the earlier batches left open:
//     ops_below / n_slots >= sample_limit * (share of k-mers still open)
// An unknown op count (a shard whose n_ops the host was not told) counts as plausible. Rank 0 has nothing below it but the earlier
// batches, which the caller's `done_before` covers: always FRONT. sample_limit 0: no k-mer ever completes: always FRONT.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

enum PgJobStatsPlace { PG_JOB_STATS_FRONT = 0, PG_JOB_STATS_BEHIND = 1 };

// shard_ops[h], shard_reads[h] for h < rank: ss ops (0 = unknown to the host) and reads of the shards below this rank
static inline bool pg_job_completion_plausible(uint32_t rank, const uint64_t *shard_ops, const uint64_t *shard_reads, uint32_t n_slots,
                                               uint64_t sample_limit, bool have_batch, uint64_t full_slots_prev) {
    uint64_t ops_below = 0; bool ops_known = true;
    for (uint32_t h = 0; h < rank; ++h) { ops_below += shard_ops[h]; if (!shard_ops[h] && shard_reads[h] > 0) ops_known = false; }
    const double ns = (double)(n_slots ? n_slots : 1);
    const double open_share = have_batch ? 1.0 - (double)full_slots_prev / ns : 1.0;
    return !ops_known || (double)ops_below / ns >= (double)sample_limit * open_share;
}

// mode: nullptr / "auto" = the rule above; "front" / "behind" = always that place for ranks > 0 (measurements, and the way out if the
// heuristic turns out wrong on real hardware)
static inline PgJobStatsPlace pg_job_stats_place(uint32_t rank, const uint64_t *shard_ops, const uint64_t *shard_reads, uint32_t n_slots, uint64_t sample_limit,
                                                 bool have_batch, uint64_t full_slots_prev, const char *mode) {
    if (rank == 0 || sample_limit == 0) return PG_JOB_STATS_FRONT;
    if (mode && !strcmp(mode, "front")) return PG_JOB_STATS_FRONT;
    if (mode && !strcmp(mode, "behind")) return PG_JOB_STATS_BEHIND;
    return pg_job_completion_plausible(rank, shard_ops, shard_reads, n_slots, sample_limit, have_batch, full_slots_prev) ? PG_JOB_STATS_BEHIND : PG_JOB_STATS_FRONT;
}
