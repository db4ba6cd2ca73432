// plan.hip - host-side launch planning for K1 (no device code).
#include "kernels.h"
#include <cstdlib>
#include <cstdio>

namespace ssmi {

static int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

bool plan_flat_scan(int dtype, int ld, long long nrows, int nq, int k, int num_cu, ScanPlan* p, std::string* err) {
    if (k < 1 || k > kMaxK) { *err = "k must be in [1, " + std::to_string(kMaxK) + "]"; return false; }
    const int V = dtype == 0 ? 4 : 8;
    const int chunk = 64 * V;
    const int NF = 4096 / chunk;
    const int nch_rt = (ld + chunk - 1) / chunk;
    const int ldq = nch_rt * chunk;
    const bool whole = (ld % chunk) == 0;
    p->r = pow2_ceil((k + 63) / 64);
    const int kp = 64 * p->r;

    int bt = pow2_ceil(nq < 1 ? 1 : nq);
    const int bt_cap = 8 / p->r < 1 ? 1 : 8 / p->r;
    if (bt > bt_cap) bt = bt_cap;
    if (bt > 8) bt = 8;
    while (bt > 1 && (size_t)bt * ldq * 4 > 128 * 1024) bt >>= 1;
    if ((size_t)ldq * 4 > 128 * 1024) { *err = "dimension too large for the scan kernel"; return false; }
    p->bt = bt;

    const char* env_nt = getenv("SSMI_SCAN_NT");
    p->nt = !(env_nt && env_nt[0] == '0');
    const char* env_qreg = getenv("SSMI_SCAN_QREG");
    const bool allow_qreg = !(env_qreg && env_qreg[0] == '0');

    const int nch = whole ? ld / chunk : 0;
    p->qreg = allow_qreg && bt == 1 && whole && (nch == NF || nch == NF / 2 || nch == NF / 4);
    int blocks_per_cu;
    if (p->qreg) {
        p->nch = nch;
        // one block per CU (12 or 8 waves): the same waves in flight as 3 small blocks, but a third of
        // the per-block partial lists for K3 to merge
        p->threads = p->r <= 4 ? 768 : 512;
        p->lds = (size_t)(p->threads / 64) * kp * sizeof(skey_t);
        blocks_per_cu = (nch == NF) ? 1 : 2;  // shorter rows need fewer registers: two blocks fit
    } else {
        p->nch = 0;
        const size_t qbytes = (size_t)bt * ldq * 4;
        p->threads = qbytes > 80 * 1024 ? 1024 : (qbytes > 40 * 1024 ? 512 : 256);
        if (const char* e = getenv("SSMI_SCAN_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) p->threads = v; }
        const size_t mbytes = (size_t)(p->threads / 64) * kp * sizeof(skey_t);
        p->lds = qbytes > mbytes ? qbytes : mbytes;
        blocks_per_cu = (int)((160 * 1024) / (p->lds + 512));
        const int wave_cap = 32 / (p->threads / 64);
        if (blocks_per_cu > wave_cap) blocks_per_cu = wave_cap;
        if (blocks_per_cu > 4) blocks_per_cu = 4;
        if (blocks_per_cu < 1) blocks_per_cu = 1;
        // long rows, 2-4 queries: one block of 8 waves per CU (8 rows of 8-16 KiB in flight) instead of 16 waves. Measured at
        // 1M x 4096 (tools/time_env_ab.py, SSMI_SCAN_THREADS x SSMI_SCAN_BPC): fp32 rows 2 queries 2.54 -> 2.35 ms, 3-4 queries
        // 2.43 -> 2.35 ms (0.87 of 8 TB/s, what one query reaches); fp16 rows 2 queries 1.335 -> 1.177 ms; fp16 rows with 3-4
        // queries (1.29 ms) and 5-8 queries on fp32 rows (2.52 ms) lose with fewer waves and keep 16
        const size_t row_bytes = (size_t)ld * (dtype == 0 ? 4 : 2);
        if (!getenv("SSMI_SCAN_THREADS") && row_bytes >= 8192 && (bt == 2 || (bt == 4 && dtype == 0)) && mbytes <= 64 * 1024) {
            p->threads = 512;
            p->lds = qbytes > (size_t)8 * kp * sizeof(skey_t) ? qbytes : (size_t)8 * kp * sizeof(skey_t);
            blocks_per_cu = 1;
        }
    }
    // K1m (scan_mm.h): four rows per wave pass, multiply-adds on the matrix pipe, queries in quads
    p->mm = false;
    p->mm_g = 0;
    {
        const char* e = getenv("SSMI_SCAN_MM");  // 0: never; 1: every shape it can take; 1a: also 1-2 queries (as a quad); unset: the policy below
        const int mm_nch = ld / (16 * V);        // 256-byte chunks per row
        int mm_g = 0;                            // loads per lane and half stage: the largest that divides half a row
        if ((ld % (16 * V)) == 0)
            for (int g = 16; g >= 1 && !mm_g; g >>= 1) if (mm_nch % (2 * g) == 0) mm_g = g;
        if (const char* eg = getenv("SSMI_MM_G")) {  // experiments
            const int v = atoi(eg);
            if ((v == 1 || v == 2 || v == 4 || v == 8 || v == 16) && mm_nch % (2 * v) == 0) mm_g = v;
        }
        const size_t row_bytes = (size_t)ld * (dtype == 0 ? 4 : 2);
        const bool short_rows = row_bytes < 8192;
        // Policy, measured at 8-16 GB of rows per shape (tools/time_env_ab.py).  Ahead of the register-FMA
        // form: 5-8 queries and 3-4 queries on 16-bit rows at every row length (1M x 4096 fp32 x 8: 2.54 -> 2.44 ms; 4M x 1024
        // fp32 x 8: 3.99 -> 2.48 ms; 4M x 512 fp16 x 4: 1.76 -> 0.79 ms); 2-4 queries on rows under 8 KiB (one or two queries run
        // as a quad: the matrix pipe has the time); one query wherever the register-query form of K1 does not apply (rows
        // other than 1/2/4/8/16 KiB: 8M x 256 fp32 1.82 -> 1.29 ms, 512 / 1536 / 3072-d +2-4 %, 768-d equal).  Behind it:
        // 2-4 queries on fp32 rows and 2 on 16-bit rows of 8 KiB and more, where the 8-wave register-FMA form holds 0.87 of the
        // HBM rate and this one 0.85.
        int mm_bt = bt;
        bool want = bt == 8 || (bt == 4 && (dtype != 0 || short_rows));
        if (bt_cap >= 4 && ((bt == 2 && short_rows) || bt == 1)) { want = true; mm_bt = 4; }
        if (e) {
            want = e[0] == '1';
            if (e[0] == '1' && e[1] == 'a' && bt < 4 && bt_cap >= 4) mm_bt = 4;
            else if (bt < 4) mm_bt = bt;
        }
        if (want && !p->qreg && mm_g > 0 && (mm_bt == 4 || mm_bt == 8) && (size_t)mm_bt * ld * 4 <= 128 * 1024) {
            p->mm = true;
            p->mm_g = mm_g;
            p->bt = bt = mm_bt;
            p->threads = mm_g <= 8 ? 1024 : 512;
            const size_t qbytes = (size_t)bt * ld * 4, mbytes = (size_t)(p->threads / 64) * kp * sizeof(skey_t);
            p->lds = qbytes > mbytes ? qbytes : mbytes;
            blocks_per_cu = 1;
        }
    }
    const char* env_bpc = getenv("SSMI_SCAN_BPC");
    if (env_bpc) { int v = atoi(env_bpc); if (v >= 1 && v <= 16) blocks_per_cu = v; }
    const int nwaves = p->threads / 64;
    long long gx = (long long)num_cu * blocks_per_cu;
    int min_rows = 1;  // rows per wave below which more blocks only add partial lists (the merge thins them cheaply)
    if (const char* e = getenv("SSMI_SCAN_MINROWS")) { const int v = atoi(e); if (v >= 1 && v <= 64) min_rows = v; }
    if (p->mm) min_rows *= 4;  // a wave pass is four rows
    const long long max_useful = (nrows + (long long)nwaves * min_rows - 1) / ((long long)nwaves * min_rows);
    if (gx > max_useful) gx = max_useful;
    if (gx < 1) gx = 1;
    p->grid_x = (int)gx;
    p->grid_y = (nq + bt - 1) / bt;
    return true;
}

}  // namespace ssmi
