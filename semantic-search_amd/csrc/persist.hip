// persist.hip - vectors.faiss: write_index / read_index in the FAISS index-file layout, and the incremental
// checkpoint of a growing Flat index.  Stands behind faiss.write_index (core/index_manager.py:241) and
// faiss.read_index (core/index_manager.py:185).
#include "index.h"
#include "kernels.h"
#include "ivf_state.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <unistd.h>

using namespace ssmi;

// ---------------------------------------------------------------------------
// persistence: FAISS index-file layout (faiss/impl/index_write.cpp [FAISS-lib],
// restated from the published format; not yet validated against a FAISS-written
// file because none exists in this environment):
//   IndexFlatIP   "IxFI" | header | n_floats:u64 | n_floats * f32
//   IndexIVFFlat  "IwFl" | header | nlist:u64 | nprobe:u64 | quantizer index ("IxFI"...) |
//                 direct map (type:u8, array: u64 n + n * i64) |
//                 "ilar" | nlist:u64 | code_size:u64 | "full" | sizes: u64 n + n * u64 |
//                 per non-empty list: codes (size * code_size bytes) then ids (size * i64)
//   header        d:i32 | ntotal:i64 | 1<<20:i64 | 1<<20:i64 | is_trained:u8 | metric_type:i32 (0 = inner product)
// Vectors are written as fp32 whatever the storage type (an fp16-storage index
// round-trips exactly: fp16 -> fp32 -> fp16 is lossless).
// ---------------------------------------------------------------------------
namespace {

struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
};
bool wr(FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
bool rd(FILE* f, void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
template <typename T> bool wr1(FILE* f, T v) { return wr(f, &v, sizeof v); }
template <typename T> bool rd1(FILE* f, T* v) { return rd(f, v, sizeof *v); }

bool write_header(FILE* f, int d, int64_t ntotal, bool trained) {
    const int64_t dummy = 1 << 20;
    return wr1<int32_t>(f, d) && wr1<int64_t>(f, ntotal) && wr1<int64_t>(f, dummy) && wr1<int64_t>(f, dummy) &&
           wr1<uint8_t>(f, trained ? 1 : 0) && wr1<int32_t>(f, 0);
}
int read_header(FILE* f, int* d, int64_t* ntotal, bool* trained) {
    int32_t dd, metric; int64_t nt, d1, d2; uint8_t tr;
    if (!(rd1(f, &dd) && rd1(f, &nt) && rd1(f, &d1) && rd1(f, &d2) && rd1(f, &tr) && rd1(f, &metric)))
        return fail("read_index: truncated index header");
    if (metric != 0) return fail("read_index: metric type %d is not supported (only METRIC_INNER_PRODUCT)", metric);
    if (dd < 1 || nt < 0) return fail("read_index: corrupt header (d=%d ntotal=%lld)", dd, (long long)nt);
    *d = dd; *ntotal = nt; *trained = tr != 0;
    return 0;
}

// rows named by `slots` of a row store -> fp32 (n, d) on the host, through the gather kernel
int read_rows(ssmi_index* h, const RowStore& R, long long view_rows, const int64_t* slots, int64_t n, float* out) {
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)R.d * 4));
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        const size_t ids_bytes = ((size_t)cnt * 8 + 255) / 256 * 256;
        CHECK(ensure_ws(h, 0, ids_bytes + 256 + (size_t)cnt * R.d * 4));
        int64_t* dids = (int64_t*)h->ws[0];
        int* bad = (int*)((char*)h->ws[0] + ids_bytes);
        float* dout = (float*)((char*)h->ws[0] + ids_bytes + 256);
        HIPCHECK(hipMemcpyAsync(dids, slots + done, (size_t)cnt * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(hipMemsetAsync(bad, 0, 4, h->stream));
        RowView v = R.view();
        v.nrows = view_rows;
        HIPCHECK(launch_gather_rows(v, R.dtype, dids, cnt, R.d, dout, bad, h->stream));
        HIPCHECK(hipMemcpyAsync(out + (size_t)done * R.d, dout, (size_t)cnt * R.d * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

int write_flat_payload(ssmi_index* h, const RowStore& R, long long nrows, FILE* f) {
    if (!wr1<uint64_t>(f, (uint64_t)nrows * (uint64_t)R.d)) return fail("write_index: write failed");
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)R.d * 4));
    std::vector<int64_t> ids((size_t)std::min<int64_t>(chunk, std::max<long long>(nrows, 1)));
    std::vector<float> buf(ids.size() * (size_t)R.d);
    for (int64_t done = 0; done < nrows; done += chunk) {
        const int64_t cnt = std::min<int64_t>(chunk, nrows - done);
        std::iota(ids.begin(), ids.begin() + cnt, done);
        CHECK(read_rows(h, R, nrows, ids.data(), cnt, buf.data()));
        if (!wr(f, buf.data(), (size_t)cnt * R.d * 4)) return fail("write_index: write failed");
    }
    return 0;
}

// ---- IVF journal (our own side file "<path>.journal"; the base file stays a plain FAISS IwFl file) -------------
//   "SSJ1" | d:i32 | nlist:u64 | base_ntotal:i64 | base_bytes:u64          header, ties the journal to ONE base file
//   "chnk" | n:u64 | labels:i32[n] | ids:i64[n] | vectors:f32[n*d]         one record per incremental checkpoint
// A record is valid only if it is complete; a torn tail (crash while appending) is ignored by the reader.
const size_t kJournalHeader = 4 + 4 + 8 + 8 + 8;
std::string journal_path(const char* path) { return std::string(path) + ".journal"; }

unsigned long long file_bytes(const char* path) {
    File F;
    F.f = fopen(path, "rb");
    if (!F.f || fseek(F.f, 0, SEEK_END) != 0) return 0;
    const long n = ftell(F.f);
    return n < 0 ? 0 : (unsigned long long)n;
}

void record_full_state(ssmi_index* h, const char* path) {
    IvfState* S = st(h);
    S->persist_path = path;
    S->persist_base_ntotal = h->ivf_ntotal;
    S->persist_base_bytes = file_bytes(path);
    S->persist_journal_rows = 0;
    S->persist_journal_bytes = 0;
    S->persist_size = S->size;
}

// does `f` (positioned at 0) start with the journal header of this index over this base file?
bool journal_header_matches(FILE* f, int d, int nlist, long long base_ntotal, unsigned long long base_bytes) {
    char cc[4];
    int32_t jd; uint64_t jnlist, jbase_bytes; int64_t jbase_nt;
    if (!(rd(f, cc, 4) && rd1(f, &jd) && rd1(f, &jnlist) && rd1(f, &jbase_nt) && rd1(f, &jbase_bytes))) return false;
    return memcmp(cc, "SSJ1", 4) == 0 && jd == d && jnlist == (uint64_t)nlist && jbase_nt == base_ntotal && jbase_bytes == base_bytes;
}

// members each list gained since the last checkpoint -> one journal record
int append_journal(ssmi_index* h, const char* path) {
    IvfState* S = st(h);
    const RowStore& R = h->rows;
    const std::string jp = journal_path(path);
    std::vector<int32_t> labels;
    std::vector<int64_t> slots;
    for (int l = 0; l < h->nlist; ++l)
        for (long long pos = S->persist_size[l]; pos < S->size[l]; ++pos) {
            labels.push_back(l);
            slots.push_back((long long)S->pages[l][pos >> kPageShift] * kPageRows + (pos & (kPageRows - 1)));
        }
    const int64_t n = (int64_t)labels.size();
    if (n == 0) return 0;
    std::vector<uint32_t> sids((size_t)h->ivf_slots_used);
    HIPCHECK(hipMemcpy(sids.data(), S->d_slot_ids, sids.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> ids64((size_t)n);
    for (int64_t i = 0; i < n; ++i) ids64[i] = (int64_t)sids[(size_t)slots[i]];
    // The record goes exactly behind the bytes this handle knows to be good (header of THIS base file + complete
    // records): a torn tail left by a crash or a full disk is cut off first, and a journal that belongs to another
    // base file (a crash between a base rewrite and the removal of its old journal) is started afresh - a record
    // written behind either would be dropped, or misread, by the next read_index.
    unsigned long long good = S->persist_journal_bytes;
    File F;
    if (good >= kJournalHeader) {
        F.f = fopen(jp.c_str(), "r+b");
        if (!F.f || file_bytes(jp.c_str()) < good ||
            !journal_header_matches(F.f, R.d, h->nlist, S->persist_base_ntotal, S->persist_base_bytes)) {
            // the journal this handle wrote is gone or was replaced behind its back: its records cannot be trusted
            // any more, so the whole index is written again
            if (F.f) { fclose(F.f); F.f = nullptr; }
            return ssmi_index_write(h, path);
        }
        if (fflush(F.f) != 0 || ftruncate(fileno(F.f), (off_t)good) != 0 || fseek(F.f, (long)good, SEEK_SET) != 0)
            return fail("write_index: cannot position %s", jp.c_str());
    } else {
        F.f = fopen(jp.c_str(), "wb");
        if (!F.f) return fail("write_index: cannot open %s", jp.c_str());
        if (!(wr(F.f, "SSJ1", 4) && wr1<int32_t>(F.f, R.d) && wr1<uint64_t>(F.f, (uint64_t)h->nlist) &&
              wr1<int64_t>(F.f, S->persist_base_ntotal) && wr1<uint64_t>(F.f, S->persist_base_bytes)) || fflush(F.f) != 0)
            return fail("write_index: journal write failed");
        good = kJournalHeader;
        S->persist_journal_bytes = good;
    }
    FILE* f = F.f;
    // a record that fails part-way is taken back, so the file never keeps half a record in front of a later one
    auto undo = [&](const char* what) -> int {
        (void)fflush(f);
        (void)!ftruncate(fileno(f), (off_t)good);
        return fail("write_index: %s for %s", what, jp.c_str());
    };
    if (!(wr(f, "chnk", 4) && wr1<uint64_t>(f, (uint64_t)n) && wr(f, labels.data(), (size_t)n * 4) && wr(f, ids64.data(), (size_t)n * 8)))
        return undo("journal write failed");
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)R.d * 4));
    std::vector<float> buf((size_t)std::min(chunk, n) * R.d);
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        if (read_rows(h, R, h->ivf_slots_used, slots.data() + done, cnt, buf.data())) { (void)fflush(f); (void)!ftruncate(fileno(f), (off_t)good); return 1; }
        if (!wr(f, buf.data(), (size_t)cnt * R.d * 4)) return undo("journal write failed");
    }
    if (fflush(f) != 0) return undo("flush failed");
    S->persist_journal_bytes = good + 12 + (unsigned long long)n * (4 + 8 + (unsigned long long)R.d * 4);
    S->persist_journal_rows += n;
    S->persist_size = S->size;
    return 0;
}

// replay "<path>.journal" into a freshly read base index; a journal of another base file is ignored
int replay_journal(ssmi_index* h, const char* path, int64_t base_ntotal) {
    IvfState* S = st(h);
    const int d = h->rows.d;
    const std::string jp = journal_path(path);
    const unsigned long long jbytes = file_bytes(jp.c_str());
    if (jbytes < kJournalHeader) return 0;
    File F;
    F.f = fopen(jp.c_str(), "rb");
    if (!F.f) return 0;
    FILE* f = F.f;
    char cc[4];
    if (!journal_header_matches(f, d, h->nlist, base_ntotal, file_bytes(path)))
        return 0;  // belongs to an earlier base file: stale, the base alone is the index (the next append rewrites it)
    unsigned long long pos = kJournalHeader;
    S->persist_journal_bytes = pos;  // header is ours; complete records extend it below, a torn tail does not
    const uint64_t rec = 4 + 8 + (uint64_t)d * 4;
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)d * 4));
    std::vector<int32_t> labels; std::vector<int64_t> ids64; std::vector<uint32_t> ids32; std::vector<float> buf; std::vector<int> lab;
    while (pos + 12 <= jbytes) {
        uint64_t n;
        if (fseek(f, (long)pos, SEEK_SET) != 0 || !rd(f, cc, 4) || memcmp(cc, "chnk", 4) != 0 || !rd1(f, &n)) break;
        if (n == 0 || n > (jbytes - pos - 12) / rec) break;  // torn tail: the complete records before it stand
        const unsigned long long lab0 = pos + 12, ids0 = lab0 + n * 4, vec0 = ids0 + n * 8;
        for (uint64_t done = 0; done < n; done += (uint64_t)chunk) {
            const int64_t cnt = (int64_t)std::min<uint64_t>((uint64_t)chunk, n - done);
            labels.resize((size_t)cnt); ids64.resize((size_t)cnt); ids32.resize((size_t)cnt); buf.resize((size_t)cnt * d); lab.resize((size_t)cnt);
            if (fseek(f, (long)(lab0 + done * 4), SEEK_SET) != 0 || !rd(f, labels.data(), (size_t)cnt * 4) ||
                fseek(f, (long)(ids0 + done * 8), SEEK_SET) != 0 || !rd(f, ids64.data(), (size_t)cnt * 8) ||
                fseek(f, (long)(vec0 + done * (uint64_t)d * 4), SEEK_SET) != 0 || !rd(f, buf.data(), (size_t)cnt * d * 4))
                return fail("read_index: journal %s is unreadable", jp.c_str());
            for (int64_t i = 0; i < cnt; ++i) {
                if (labels[i] < 0 || labels[i] >= h->nlist || ids64[i] < 0 || ids64[i] >= 0xFFFFFFFFll)
                    return fail("read_index: corrupt journal record in %s", jp.c_str());
                lab[i] = labels[i]; ids32[i] = (uint32_t)ids64[i];
            }
            const float* xdev = nullptr;
            CHECK(rows_to_device_f32(h, cnt, buf.data(), SSMI_F32, SSMI_HOST, &xdev));
            CHECK(ivf_append_preassigned(h, cnt, xdev, lab.data(), ids32.data()));
        }
        S->persist_journal_rows += (long long)n;
        pos = vec0 + n * (uint64_t)d * 4;
        S->persist_journal_bytes = pos;
    }
    return 0;
}

}  // namespace
extern "C" {

int ssmi_index_write(ssmi_index* h, const char* path) {
    return guarded([&]() -> int {
    if (!h || !path) return fail("NULL argument");
    DeviceGuard g(h->device);
    quiesce(h);
    File F;
    F.f = fopen(path, "wb");
    if (!F.f) return fail("write_index: cannot open %s for writing", path);
    FILE* f = F.f;
    const RowStore& R = h->rows;
    if (h->kind == SSMI_KIND_FLAT) {
        if (!wr(f, "IxFI", 4) || !write_header(f, R.d, R.nrows, true)) return fail("write_index: write failed");
        CHECK(write_flat_payload(h, R, R.nrows, f));
    } else {
        IvfState* S = st(h);
        if (!wr(f, "IwFl", 4) || !write_header(f, R.d, h->ivf_ntotal, h->trained)) return fail("write_index: write failed");
        if (!wr1<uint64_t>(f, (uint64_t)h->nlist) || !wr1<uint64_t>(f, (uint64_t)h->nprobe)) return fail("write_index: write failed");
        const long long ncent = h->trained ? h->nlist : 0;
        if (!wr(f, "IxFI", 4) || !write_header(f, R.d, ncent, true)) return fail("write_index: write failed");
        CHECK(write_flat_payload(h, S->cent32, ncent, f));
        if (!wr1<uint8_t>(f, 0) || !wr1<uint64_t>(f, 0)) return fail("write_index: write failed");  // direct map: none
        if (!wr(f, "ilar", 4) || !wr1<uint64_t>(f, (uint64_t)h->nlist) || !wr1<uint64_t>(f, (uint64_t)R.d * 4) ||
            !wr(f, "full", 4) || !wr1<uint64_t>(f, (uint64_t)h->nlist))
            return fail("write_index: write failed");
        for (int l = 0; l < h->nlist; ++l)
            if (!wr1<uint64_t>(f, (uint64_t)S->size[l])) return fail("write_index: write failed");
        std::vector<uint32_t> sids((size_t)h->ivf_slots_used);
        if (!sids.empty()) HIPCHECK(hipMemcpy(sids.data(), S->d_slot_ids, sids.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int64_t> slots, ids64;
        std::vector<float> buf;
        for (int l = 0; l < h->nlist; ++l) {
            const long long n = S->size[l];
            if (n == 0) continue;
            slots.resize((size_t)n); ids64.resize((size_t)n); buf.resize((size_t)n * R.d);
            for (long long pos = 0; pos < n; ++pos) {
                slots[pos] = (long long)S->pages[l][pos >> kPageShift] * kPageRows + (pos & (kPageRows - 1));
                ids64[pos] = (int64_t)sids[(size_t)slots[pos]];
            }
            CHECK(read_rows(h, R, h->ivf_slots_used, slots.data(), n, buf.data()));
            if (!wr(f, buf.data(), (size_t)n * R.d * 4) || !wr(f, ids64.data(), (size_t)n * 8)) return fail("write_index: write failed");
        }
    }
    if (fflush(f) != 0) return fail("write_index: flush failed for %s", path);
    if (h->kind == SSMI_KIND_IVFFLAT) {
        fclose(F.f); F.f = nullptr;
        remove(journal_path(path).c_str());  // a full write supersedes the journal of the previous base file
        record_full_state(h, path);
    }
    return 0;
    });
}

// Checkpoint of a growing Flat index: when `path` already holds an earlier state of THIS index (same d, a
// prefix of its rows - checked on the header, the file size and the last stored row), only the rows added since
// are appended and the two counts in the header are patched; the result is byte-identical to ssmi_index_write.
// Anything else (no file, another index, an IVF index) is a full rewrite.  The reference rewrites the whole file
// at every checkpoint (core/index_manager.py:859-860 -> :241): N*d*4 bytes each time.
int ssmi_index_write_append(ssmi_index* h, const char* path) {
    return guarded([&]() -> int {
    if (!h || !path) return fail("NULL argument");
    DeviceGuard g(h->device);
    quiesce(h);
    if (h->kind != SSMI_KIND_FLAT) {
        // IVF: the members gained since the last checkpoint go to "<path>.journal" (one record); the base file is
        // rewritten only when this handle has no full state for `path`, the base file changed under it, or the
        // journal has grown to half the base (then the next read would replay too much)
        IvfState* S = st(h);
        const bool same = S->persist_path == path && S->persist_base_ntotal >= 0 && (int)S->persist_size.size() == h->nlist &&
                          file_bytes(path) == S->persist_base_bytes && h->ivf_ntotal >= S->persist_base_ntotal + S->persist_journal_rows;
        const long long fresh_rows = h->ivf_ntotal - (same ? S->persist_base_ntotal + S->persist_journal_rows : 0);
        if (!same || S->persist_journal_rows + fresh_rows > S->persist_base_ntotal / 2 + 65536) return ssmi_index_write(h, path);
        return append_journal(h, path);
    }
    const RowStore& R = h->rows;
    const long kHeader = 4 + 33;  // fourcc + header; then n_floats:u64, then the rows
    int64_t have = -1;
    {
        File F;
        F.f = fopen(path, "rb");
        if (F.f) {
            char cc[4];
            int d = 0; int64_t nt = 0; bool tr = false; uint64_t nfl = 0;
            if (rd(F.f, cc, 4) && memcmp(cc, "IxFI", 4) == 0 && read_header(F.f, &d, &nt, &tr) == 0 && rd1(F.f, &nfl) &&
                d == R.d && nt <= R.nrows && nfl == (uint64_t)nt * (uint64_t)d && fseek(F.f, 0, SEEK_END) == 0 &&
                (uint64_t)ftell(F.f) == (uint64_t)kHeader + 8 + nfl * 4) {
                have = nt;
                if (nt > 0) {  // the last stored row must be row nt-1 of this index
                    std::vector<float> on_disk((size_t)d), ours((size_t)d);
                    const int64_t id = nt - 1;
                    if (fseek(F.f, kHeader + 8 + (long)((uint64_t)id * d * 4), SEEK_SET) != 0 || !rd(F.f, on_disk.data(), (size_t)d * 4)) have = -1;
                    else {
                        CHECK(read_rows(h, R, R.nrows, &id, 1, ours.data()));
                        if (memcmp(on_disk.data(), ours.data(), (size_t)d * 4) != 0) have = -1;
                    }
                }
            }
        }
    }
    if (have < 0) return ssmi_index_write(h, path);
    if (have == R.nrows) return 0;  // nothing new
    File F;
    F.f = fopen(path, "r+b");
    if (!F.f) return ssmi_index_write(h, path);
    FILE* f = F.f;
    if (fseek(f, 0, SEEK_END) != 0) return fail("write_index: seek failed for %s", path);
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)R.d * 4));
    std::vector<int64_t> ids((size_t)std::min<int64_t>(chunk, R.nrows - have));
    std::vector<float> buf(ids.size() * (size_t)R.d);
    for (int64_t done = have; done < R.nrows; done += chunk) {
        const int64_t cnt = std::min<int64_t>(chunk, R.nrows - done);
        std::iota(ids.begin(), ids.begin() + cnt, done);
        CHECK(read_rows(h, R, R.nrows, ids.data(), cnt, buf.data()));
        if (!wr(f, buf.data(), (size_t)cnt * R.d * 4)) return fail("write_index: write failed");
    }
    // rows first, counts last: a crash in between leaves the old, still valid index with a longer tail
    if (fflush(f) != 0) return fail("write_index: flush failed for %s", path);
    if (fseek(f, 4 + 4, SEEK_SET) != 0 || !wr1<int64_t>(f, (int64_t)R.nrows) || fseek(f, kHeader, SEEK_SET) != 0 ||
        !wr1<uint64_t>(f, (uint64_t)R.nrows * (uint64_t)R.d) || fflush(f) != 0)
        return fail("write_index: header update failed for %s", path);
    return 0;
    });
}

// reads "IxFI" + header + payload into an existing empty flat-like store via store_add
static int read_flat_into(ssmi_index* h, RowStore& R, FILE* f, int expect_d, int64_t* nread) {
    char cc[4];
    if (!rd(f, cc, 4)) return fail("read_index: truncated file");
    if (memcmp(cc, "IxFI", 4) != 0) return fail("read_index: unsupported index type '%.4s' (expected IxFI)", cc);
    int d; int64_t nt; bool tr;
    CHECK(read_header(f, &d, &nt, &tr));
    if (expect_d > 0 && d != expect_d) return fail("read_index: nested index dimension %d != %d", d, expect_d);
    uint64_t nfl;
    if (!rd1(f, &nfl)) return fail("read_index: truncated file");
    if (nfl != (uint64_t)nt * (uint64_t)d) return fail("read_index: payload size %llu != ntotal*d", (unsigned long long)nfl);
    {
        const long here = ftell(f);
        if (here < 0 || fseek(f, 0, SEEK_END) != 0) return fail("read_index: seek failed");
        const long end = ftell(f);
        if (fseek(f, here, SEEK_SET) != 0) return fail("read_index: seek failed");
        if (end < here || (uint64_t)(end - here) / 4 < nfl) return fail("read_index: truncated vector payload (%llu floats claimed, %llu bytes left)", (unsigned long long)nfl, (unsigned long long)(end - here));
    }
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)d * 4));
    std::vector<float> buf((size_t)std::min<int64_t>(chunk, std::max<int64_t>(nt, 1)) * d);
    for (int64_t done = 0; done < nt; done += chunk) {
        const int64_t cnt = std::min(chunk, nt - done);
        if (!rd(f, buf.data(), (size_t)cnt * d * 4)) return fail("read_index: truncated vector payload");
        CHECK(store_add(h, R, cnt, buf.data(), SSMI_F32, SSMI_HOST));
    }
    *nread = nt;
    return 0;
}

// members of one inverted list, in stored order (ids, optionally vectors as fp32 on the host)
int ssmi_index_get_list_members(ssmi_index* h, int list, int64_t cap, int64_t* ids_out_host, float* vecs_out_host, int64_t* n_out) {
    return guarded([&]() -> int {
    if (!h || !n_out) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("get_list_members: not an IVF index");
    if (list < 0 || list >= h->nlist) return fail("get_list_members: list %d out of range [0, %d)", list, h->nlist);
    DeviceGuard g(h->device);
    quiesce(h);
    IvfState* S = st(h);
    const long long n = S->size[list];
    *n_out = n;
    if (n == 0 || n > cap) return 0;
    if (!ids_out_host) return fail("NULL buffer");
    std::vector<int64_t> slots((size_t)n);
    for (long long pos = 0; pos < n; ++pos)
        slots[pos] = (long long)S->pages[list][pos >> kPageShift] * kPageRows + (pos & (kPageRows - 1));
    // the ids of the list's slots: every page is a run of kPageRows slot ids
    std::vector<uint32_t> page_ids((size_t)(n + kPageRows - 1) / kPageRows * kPageRows);
    for (long long pg = 0; pg * kPageRows < n; ++pg)
        HIPCHECK(hipMemcpyAsync(page_ids.data() + pg * kPageRows, S->d_slot_ids + (size_t)S->pages[list][pg] * kPageRows,
                                (size_t)std::min<long long>(kPageRows, n - pg * kPageRows) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (long long j = 0; j < n; ++j) ids_out_host[j] = (int64_t)page_ids[(size_t)j];
    if (vecs_out_host) CHECK(read_rows(h, h->rows, h->ivf_slots_used, slots.data(), n, vecs_out_host));
    return 0;
    });
}

int ssmi_index_read(const char* path, int storage_dtype, int device, ssmi_index** out) {
    return guarded([&]() -> int {
    if (!path || !out) return fail("NULL argument");
    *out = nullptr;
    if (storage_dtype < 0) storage_dtype = SSMI_F32;
    File F;
    F.f = fopen(path, "rb");
    if (!F.f) return fail("read_index: cannot open %s", path);
    FILE* f = F.f;
    char cc[4];
    if (!rd(f, cc, 4)) return fail("read_index: %s is empty", path);
    if (memcmp(cc, "IxFI", 4) == 0) {
        // peek the dimension, then let read_flat_into re-parse from the start
        int32_t d;
        if (!rd1(f, &d) || d < 1) return fail("read_index: corrupt header");
        if (fseek(f, 0, SEEK_SET) != 0) return fail("read_index: seek failed");
        ssmi_index* h = nullptr;
        CHECK(ssmi_index_create_flat(d, storage_dtype, device, &h));
        DeviceGuard g(device);
        int64_t n;
        if (read_flat_into(h, h->rows, f, d, &n)) { ssmi_index_free(h); return 1; }
        *out = h;
        return 0;
    }
    if (memcmp(cc, "IwFl", 4) != 0)
        return fail("read_index: unsupported index type '%.4s' (this engine reads IxFI and IwFl)", cc);
    int d; int64_t ntotal; bool trained;
    CHECK(read_header(f, &d, &ntotal, &trained));
    uint64_t nlist, nprobe;
    // nprobe above nlist is legal in a FAISS file (the search clamps it); one that does not fit an int is damage
    if (!rd1(f, &nlist) || !rd1(f, &nprobe) || nlist < 1 || nlist > (1u << 24) || nprobe > 0x7fffffffull) return fail("read_index: corrupt IVF header");
    ssmi_index* h = nullptr;
    CHECK(ssmi_index_create_ivfflat(d, (int)nlist, storage_dtype, device, &h));
    struct Guard { ssmi_index* h; ~Guard() { if (h) ssmi_index_free(h); } } guard{h};
    DeviceGuard g(device);
    h->nprobe = (int)std::max<uint64_t>(1, nprobe);
    IvfState* S = st(h);
    int64_t ncent = 0;
    S->cent32.nrows = 0;
    CHECK(read_flat_into(h, S->cent32, f, d, &ncent));
    if (ncent != 0 && ncent != (int64_t)nlist) return fail("read_index: quantizer holds %lld centroids, nlist is %llu", (long long)ncent, (unsigned long long)nlist);
    if (ncent) {
        std::vector<float> c((size_t)nlist * d);
        h->trained = true;
        CHECK(ssmi_index_get_centroids(h, c.data()));
        CHECK(install_centroids(h, c.data(), SSMI_HOST));
    }
    h->trained = trained && ncent > 0;
    uint8_t dm_type; uint64_t dm_n;
    if (!rd1(f, &dm_type) || !rd1(f, &dm_n)) return fail("read_index: truncated direct map");
    if (dm_type > 1) return fail("read_index: hashtable direct maps are not supported");
    if (dm_n > ((uint64_t)1 << 40) || fseek(f, (long)(dm_n * 8), SEEK_CUR) != 0) return fail("read_index: corrupt direct map");
    uint64_t il_nlist, code_size, nsz;
    if (!rd(f, cc, 4) || memcmp(cc, "ilar", 4) != 0) return fail("read_index: unsupported inverted-list type '%.4s'", cc);
    if (!rd1(f, &il_nlist) || !rd1(f, &code_size) || il_nlist != nlist || code_size != (uint64_t)d * 4)
        return fail("read_index: inverted lists do not match the index header");
    if (!rd(f, cc, 4)) return fail("read_index: truncated file");
    std::vector<uint64_t> sizes((size_t)nlist, 0);
    if (memcmp(cc, "full", 4) == 0) {
        if (!rd1(f, &nsz) || nsz != nlist || !rd(f, sizes.data(), (size_t)nlist * 8)) return fail("read_index: corrupt list sizes");
    } else if (memcmp(cc, "sprs", 4) == 0) {
        if (!rd1(f, &nsz) || (nsz & 1) || nsz > 2 * nlist) return fail("read_index: corrupt sparse list sizes");
        std::vector<uint64_t> pairs((size_t)nsz);
        if (!rd(f, pairs.data(), (size_t)nsz * 8)) return fail("read_index: corrupt sparse list sizes");
        for (size_t i = 0; i + 1 < pairs.size(); i += 2) {
            if (pairs[i] >= nlist) return fail("read_index: corrupt sparse list sizes");
            sizes[(size_t)pairs[i]] = pairs[i + 1];
        }
    } else {
        return fail("read_index: unknown list-size encoding '%.4s'", cc);
    }
    // nothing is allocated from a size field before the sizes have been checked against the header and against the
    // bytes the file still holds (a corrupt or truncated file must fail with a message, not with bad_alloc)
    const long list_start = ftell(f);
    if (list_start < 0 || fseek(f, 0, SEEK_END) != 0) return fail("read_index: seek failed");
    const long file_end = ftell(f);
    if (file_end < list_start || fseek(f, list_start, SEEK_SET) != 0) return fail("read_index: seek failed");
    const uint64_t bytes_left = (uint64_t)(file_end - list_start);
    const uint64_t rec = (uint64_t)d * 4 + 8;  // one member: its code and its id
    uint64_t members = 0;
    for (uint64_t l = 0; l < nlist; ++l) {
        if (sizes[(size_t)l] > bytes_left / rec) return fail("read_index: list %llu claims %llu vectors, more than the file holds", (unsigned long long)l, (unsigned long long)sizes[(size_t)l]);
        members += sizes[(size_t)l];
    }
    if (members != (uint64_t)ntotal) return fail("read_index: lists hold %llu vectors, header says %lld", (unsigned long long)members, (long long)ntotal);
    if (members * rec > bytes_left) return fail("read_index: truncated inverted lists (%llu bytes needed, %llu left)", (unsigned long long)(members * rec), (unsigned long long)bytes_left);
    // each list is read in bounded chunks: codes of the chunk, then (seek) its ids, so host memory stays at ~64 MiB
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)d * 4));
    std::vector<float> buf; std::vector<int64_t> ids64; std::vector<uint32_t> ids32; std::vector<int> labels;
    int64_t got = 0;
    long pos = list_start;
    for (uint64_t l = 0; l < nlist; ++l) {
        const int64_t n = (int64_t)sizes[(size_t)l];
        if (n == 0) continue;
        const long codes0 = pos, ids0 = pos + (long)((uint64_t)n * d * 4);
        for (int64_t done = 0; done < n; done += chunk) {
            const int64_t cnt = std::min(chunk, n - done);
            buf.resize((size_t)cnt * d); ids64.resize((size_t)cnt); ids32.resize((size_t)cnt); labels.assign((size_t)cnt, (int)l);
            if (fseek(f, codes0 + (long)((uint64_t)done * d * 4), SEEK_SET) != 0 || !rd(f, buf.data(), (size_t)cnt * d * 4) ||
                fseek(f, ids0 + (long)((uint64_t)done * 8), SEEK_SET) != 0 || !rd(f, ids64.data(), (size_t)cnt * 8))
                return fail("read_index: truncated inverted list %llu", (unsigned long long)l);
            for (int64_t i = 0; i < cnt; ++i) {
                if (ids64[i] < 0 || ids64[i] >= 0xFFFFFFFFll) return fail("read_index: id %lld out of the supported range", (long long)ids64[i]);
                ids32[i] = (uint32_t)ids64[i];
            }
            const float* xdev = nullptr;
            CHECK(rows_to_device_f32(h, cnt, buf.data(), SSMI_F32, SSMI_HOST, &xdev));
            CHECK(ivf_append_preassigned(h, cnt, xdev, labels.data(), ids32.data()));
        }
        pos = ids0 + (long)((uint64_t)n * 8);
        got += n;
    }
    if (got != ntotal) return fail("read_index: lists hold %lld vectors, header says %lld", (long long)got, (long long)ntotal);
    fclose(F.f); F.f = nullptr;
    S->persist_path = path;
    S->persist_base_ntotal = ntotal;
    S->persist_base_bytes = file_bytes(path);
    S->persist_journal_rows = 0;
    S->persist_journal_bytes = 0;
    CHECK(replay_journal(h, path, ntotal));
    S->persist_size = S->size;
    guard.h = nullptr;
    *out = h;
    return 0;
    });
}

}  // extern "C"
