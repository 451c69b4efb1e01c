// kmer_counts.cpp — see kmer_counts.hpp.
#include "kmer_counts.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <string_view>
#include <thread>

namespace pangenie {

namespace {
inline int base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}
inline uint64_t mix64(uint64_t x) {   // (splitmix64 finaliser: 2-bit codes of similar k-mers differ in few bits)
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
// where a code starts probing in a table of ANY size: the high half of hash x size (no power of two needed, so a table
// of billions of slots is sized by its load, not rounded up to twice that)
inline size_t slot_of(uint64_t code, size_t cap) { return (size_t)(((unsigned __int128)mix64(code) * (unsigned __int128)cap) >> 64); }
}  // namespace

// ------------------------------------------------------------------ ExactKmerCounter
ExactKmerCounter::ExactKmerCounter(const std::string& readfile, size_t kmer_size) : k_(kmer_size) {
    if (k_ == 0 || k_ > 32) throw std::runtime_error("ExactKmerCounter: k-mer size must be 1..32");
    std::ifstream in(readfile);
    if (!in.good()) throw std::runtime_error("ExactKmerCounter: cannot open " + readfile);
    {   // a file of n letters holds fewer than n windows: start with room for them (up to 2^27 slots; growth goes on from there)
        in.seekg(0, std::ios::end);
        const std::streamoff letters = in.tellg();
        in.seekg(0, std::ios::beg);
        size_t cap = (size_t)1 << 16;
        while (cap < ((size_t)1 << 27) && letters > 0 && (double)cap * 0.6 < (double)letters) cap <<= 1;
        keys_.assign(cap, kFree);
        seen_.assign(cap, 0);
    }
    // FASTA (">" header, sequence on one or more lines) or FASTQ ("@" header, sequence, "+", qualities)
    std::string line, seq;
    enum { NONE, FASTA, FQ_SEQ, FQ_PLUS, FQ_QUAL } state = NONE;
    size_t qual_left = 0;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (state == FQ_QUAL) {
            qual_left = line.size() >= qual_left ? 0 : qual_left - line.size();
            if (qual_left == 0) state = NONE;
            continue;
        }
        if (state == FQ_SEQ && !line.empty() && line[0] == '+') {
            add_sequence(seq);
            qual_left = seq.size();
            seq.clear();
            state = qual_left ? FQ_QUAL : NONE;
            continue;
        }
        if (!line.empty() && line[0] == '>' && state != FQ_SEQ) {
            if (state == FASTA) add_sequence(seq);
            seq.clear();
            state = FASTA;
            continue;
        }
        if (!line.empty() && line[0] == '@' && (state == NONE || state == FASTA)) {
            if (state == FASTA) add_sequence(seq);
            seq.clear();
            state = FQ_SEQ;
            continue;
        }
        if (state == FASTA || state == FQ_SEQ) seq += line;
    }
    if (state == FASTA || state == FQ_SEQ) add_sequence(seq);
}

void ExactKmerCounter::add_sequence(const std::string& seq) {
    // rolling 2-bit codes of the window and of its reverse complement; a letter outside {A,C,G,T} restarts the window.
    // The codes are counted a batch behind: the slot of each is touched when the code is formed, so the table's cache misses
    // of one batch overlap instead of following each other.
    const uint64_t mask = k_ == 32 ? ~0ull : ((1ull << (2 * k_)) - 1ull);
    constexpr size_t kBatch = 32;
    uint64_t batch[kBatch];
    size_t waiting = 0;
    uint64_t fwd = 0, rev = 0;
    size_t filled = 0;
    if (keys_.empty()) grow();
    for (char c : seq) {
        const int b = base_code(c);
        if (b < 0) { filled = 0; fwd = rev = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)b) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
        if (++filled >= k_) {
            const uint64_t code = fwd < rev ? fwd : rev;   // (2-bit codes order like the letters: A < C < G < T)
            __builtin_prefetch(&keys_[(size_t)mix64(code) & (keys_.size() - 1)]);
            batch[waiting++] = code;
            if (waiting == kBatch) { for (size_t i = 0; i < kBatch; ++i) bump(batch[i]); waiting = 0; }
        }
    }
    for (size_t i = 0; i < waiting; ++i) bump(batch[i]);
}

void ExactKmerCounter::grow() {
    const size_t cap = keys_.empty() ? (size_t)1 << 16 : keys_.size() * 2;
    std::vector<uint64_t> keys(cap, kFree);
    std::vector<uint32_t> seen(cap, 0);
    for (size_t at = 0; at < keys_.size(); ++at) {
        if (keys_[at] == kFree) continue;
        size_t to = (size_t)mix64(keys_[at]) & (cap - 1);
        while (keys[to] != kFree) to = (to + 1) & (cap - 1);
        keys[to] = keys_[at];
        seen[to] = seen_[at];
    }
    keys_.swap(keys);
    seen_.swap(seen);
}

void ExactKmerCounter::bump(uint64_t code) {
    if ((filled_ + 1) * 5 > keys_.size() * 3) grow();
    const size_t cap = keys_.size();
    size_t at = (size_t)mix64(code) & (cap - 1);
    while (keys_[at] != code) {
        if (keys_[at] == kFree) { keys_[at] = code; filled_ += 1; break; }
        at = (at + 1) & (cap - 1);
    }
    if (seen_[at] != ~0u) seen_[at] += 1;
}

bool ExactKmerCounter::encode_canonical(const char* s, uint64_t& code) const {
    uint64_t fwd = 0, rev = 0;
    for (size_t i = 0; i < k_; ++i) {
        const int b = base_code(s[i]);
        if (b < 0) return false;
        fwd = (fwd << 2) | (uint64_t)b;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
    }
    code = fwd < rev ? fwd : rev;
    return true;
}

size_t ExactKmerCounter::getKmerAbundance(std::string kmer) {
    if (kmer.size() != k_) throw std::runtime_error("ExactKmerCounter::getKmerAbundance: k-mer of length " + std::to_string(kmer.size()) + ", counter holds " + std::to_string(k_) + "-mers");
    uint64_t code;
    if (!encode_canonical(kmer.data(), code)) return 0;
    if (keys_.empty()) return 0;
    const size_t cap = keys_.size();
    for (size_t at = (size_t)mix64(code) & (cap - 1); keys_[at] != kFree; at = (at + 1) & (cap - 1))
        if (keys_[at] == code) return (size_t)seen_[at];
    return 0;
}

// ------------------------------------------------------------------ TargetedKmerCounter
namespace {
// The record grammar of FASTA / FASTQ as a machine fed line by line — the same as ExactKmerCounter's constructor: `sink` gets
// the letters of one record at a time.
struct RecordLines {
    std::string seq;
    enum { NONE, FASTA, FQ_SEQ, FQ_QUAL } state = NONE;
    size_t qual_left = 0;
    template <class Sink>
    void line(std::string_view l, Sink&& sink) {
        if (!l.empty() && l.back() == '\r') l.remove_suffix(1);
        if (state == FQ_QUAL) {
            qual_left = l.size() >= qual_left ? 0 : qual_left - l.size();
            if (qual_left == 0) state = NONE;
            return;
        }
        if (state == FQ_SEQ && !l.empty() && l[0] == '+') {
            sink(seq);
            qual_left = seq.size();
            seq.clear();
            state = qual_left ? FQ_QUAL : NONE;
            return;
        }
        if (!l.empty() && l[0] == '>' && state != FQ_SEQ) {
            if (state == FASTA) sink(seq);
            seq.clear();
            state = FASTA;
            return;
        }
        if (!l.empty() && l[0] == '@' && (state == NONE || state == FASTA)) {
            if (state == FASTA) sink(seq);
            seq.clear();
            state = FQ_SEQ;
            return;
        }
        if (state == FASTA || state == FQ_SEQ) seq.append(l.data(), l.size());
    }
    template <class Sink>
    void finish(Sink&& sink) {
        if (state == FASTA || state == FQ_SEQ) sink(seq);
        seq.clear();
        state = NONE;
    }
};

// Sequences of a FASTA / FASTQ file, plain or gzipped (zlib reads both), handed to `sink` one record at a time, over a line
// reader that never holds more than one record.
template <class Sink>
void stream_sequences(const std::string& path, Sink&& sink) {
    gzFile in = gzopen(path.c_str(), "rb");
    if (!in) throw std::runtime_error("TargetedKmerCounter: cannot open " + path);
    gzbuffer(in, 1u << 20);
    RecordLines records;
    auto handle = [&](std::string_view l) { records.line(l, sink); };
    // blocks of 4 MB, lines split in place; the unfinished tail of a block moves to the front of the next one
    std::vector<char> buf(4u << 20);
    size_t have = 0;
    try {
        while (true) {
            if (have == buf.size()) buf.resize(buf.size() * 2);   // a single line longer than the block
            const int got = gzread(in, buf.data() + have, (unsigned)(buf.size() - have));
            if (got < 0) throw std::runtime_error("TargetedKmerCounter: read error in " + path);
            const size_t end = have + (size_t)got;
            size_t at = 0;
            while (true) {
                const char* nl = (const char*)std::memchr(buf.data() + at, '\n', end - at);
                if (!nl) break;
                handle(std::string_view(buf.data() + at, (size_t)(nl - (buf.data() + at))));
                at = (size_t)(nl - buf.data()) + 1;
            }
            have = end - at;
            if (have && at) std::memmove(buf.data(), buf.data() + at, have);
            if (got == 0) break;
        }
        if (have) handle(std::string_view(buf.data(), have));   // last line without a newline
        records.finish(sink);
    } catch (...) {
        gzclose(in);
        throw;
    }
    gzclose(in);
}

}  // namespace

TargetedKmerCounter::TargetedKmerCounter(size_t kmer_size, bool unregistered_counts_zero) : k_(kmer_size), lenient_(unregistered_counts_zero) {
    if (k_ == 0 || k_ > 32) throw std::runtime_error("TargetedKmerCounter: k-mer size must be 1..32");
}

bool TargetedKmerCounter::encode_canonical(const char* s, uint64_t& code) const {
    uint64_t fwd = 0, rev = 0;
    for (size_t i = 0; i < k_; ++i) {
        const int b = base_code(s[i]);
        if (b < 0) return false;
        fwd = (fwd << 2) | (uint64_t)b;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
    }
    code = fwd < rev ? fwd : rev;
    return true;
}

void TargetedKmerCounter::add_target(std::string_view kmer) {
    if (frozen_) throw std::runtime_error("TargetedKmerCounter: targets must be registered before the reads are counted");
    if (kmer.size() != k_) throw std::runtime_error("TargetedKmerCounter::add_target: k-mer of length " + std::to_string(kmer.size()) + ", counter holds " + std::to_string(k_) + "-mers");
    uint64_t code;
    if (encode_canonical(kmer.data(), code)) pending_.push_back(code);
}

size_t TargetedKmerCounter::add_targets_from_table(const std::string& kmers_tsv_gz) {
    gzFile file = gzopen(kmers_tsv_gz.c_str(), "rb");
    if (!file) throw std::runtime_error("TargetedKmerCounter: kmer file cannot be opened.");
    std::vector<char> buf(1u << 16);
    std::string line;
    size_t rows = 0;
    try {
        while (gzgets(file, buf.data(), (int)buf.size()) != nullptr) {
            line += buf.data();
            if (line.empty() || line.back() != '\n') continue;
            line.pop_back();
            std::string chrom; size_t start = 0; std::vector<std::string> kmers, flanking; bool header = false;
            parse_kmer_line(line, chrom, start, kmers, flanking, header);
            if (!header) {
                for (const std::string& k : kmers) add_target(k);
                for (const std::string& k : flanking) add_target(k);
                rows += 1;
            }
            line.clear();
        }
    } catch (...) {
        gzclose(file);
        throw;
    }
    gzclose(file);
    return rows;
}

void TargetedKmerCounter::add_targets_of(std::string_view sequence) {
    if (frozen_) throw std::runtime_error("TargetedKmerCounter: targets must be registered before the reads are counted");
    const uint64_t mask = k_ == 32 ? ~0ull : ((1ull << (2 * k_)) - 1ull);
    uint64_t fwd = 0, rev = 0;
    size_t filled = 0;
    for (const char c : sequence) {
        const int b = base_code(c);
        if (b < 0) { filled = 0; fwd = rev = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)b) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
        if (++filled >= k_) pending_.push_back(fwd < rev ? fwd : rev);
    }
}

size_t TargetedKmerCounter::add_targets_from_sequences(const std::string& fasta) {
    if (frozen_) throw std::runtime_error("TargetedKmerCounter: targets must be registered before the reads are counted");
    const uint64_t mask = k_ == 32 ? ~0ull : ((1ull << (2 * k_)) - 1ull);
    size_t registered = 0, dedup_at = (size_t)64 << 20;
    stream_sequences(fasta, [&](const std::string& seq) {
        uint64_t fwd = 0, rev = 0;
        size_t filled = 0;
        for (const char c : seq) {
            const int b = base_code(c);
            if (b < 0) { filled = 0; fwd = rev = 0; continue; }
            fwd = ((fwd << 2) | (uint64_t)b) & mask;
            rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
            if (++filled >= k_) { pending_.push_back(fwd < rev ? fwd : rev); registered += 1; }
        }
        if (pending_.size() > dedup_at) {   // (long graphs: drop repeats now and then — each time at twice what was left)
            std::sort(pending_.begin(), pending_.end());
            pending_.erase(std::unique(pending_.begin(), pending_.end()), pending_.end());
            dedup_at = std::max<size_t>(dedup_at, 2 * pending_.size());
        }
    });
    return registered;
}

std::vector<size_t> TargetedKmerCounter::abundance_histogram(size_t max_count) {
    freeze();
    std::vector<size_t> seen(max_count + 1, 0);
    for (const Slot& slot : slots_)
        if (slot.key != kEmpty && slot.count > 0 && slot.count <= max_count) seen[(size_t)slot.count] += 1;
    return seen;
}

void TargetedKmerCounter::freeze(unsigned threads) {
    if (frozen_) return;
    // the table has two slots for every registered code (repeats included: the distinct ones are not known yet; measured:
    // 1.5 slots cost 20-40 % more time in the probes) and is of exactly that size — slot_of() needs no power of two;
    // the codes go in with compare-and-swap on the key — insert-only linear probing needs nothing more —, a batch of
    // prefetched slots at a time, from `threads` workers
    const size_t cap = std::max<size_t>(16, 2 * pending_.size() + 1);   // (any size: slot_of() scales the hash)
    slots_.resize(cap);
    std::atomic<size_t> distinct{0}, next{0}, next_clear{0};
    auto clear = [&] {
        constexpr size_t kPiece = 1u << 18;
        for (size_t from = next_clear.fetch_add(kPiece); from < cap; from = next_clear.fetch_add(kPiece))
            std::fill(slots_.begin() + (std::ptrdiff_t)from, slots_.begin() + (std::ptrdiff_t)std::min(from + kPiece, cap), Slot{kEmpty, 0});
    };
    if (threads <= 1 || cap < (1u << 20)) clear();
    else {
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < threads; ++t) workers.emplace_back(clear);
        for (std::thread& w : workers) w.join();
    }
    auto work = [&] {
        constexpr size_t kChunk = 1u << 16, kBatch = 16;
        size_t mine = 0;
        for (size_t from = next.fetch_add(kChunk); from < pending_.size(); from = next.fetch_add(kChunk)) {
            const size_t to = std::min(from + kChunk, pending_.size());
            for (size_t b = from; b < to; b += kBatch) {
                const size_t e = std::min(b + kBatch, to);
                for (size_t i = b; i < e; ++i) __builtin_prefetch(&slots_[slot_of(pending_[i], cap)], 1);
                for (size_t i = b; i < e; ++i) {
                    const uint64_t code = pending_[i];
                    size_t at = slot_of(code, cap);
                    while (true) {
                        uint64_t seen = __atomic_load_n(&slots_[at].key, __ATOMIC_RELAXED);
                        if (seen == code) break;
                        if (seen == kEmpty) {
                            if (__atomic_compare_exchange_n(&slots_[at].key, &seen, code, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { mine += 1; break; }
                            if (seen == code) break;   // (another worker put the same code here first)
                        }
                        if (++at == cap) at = 0;
                    }
                }
            }
        }
        distinct += mine;
    };
    if (threads <= 1 || pending_.size() < (1u << 18)) work();
    else {
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < threads; ++t) workers.emplace_back(work);
        for (std::thread& w : workers) w.join();
    }
    n_targets_ = distinct;
    std::vector<uint64_t>().swap(pending_);
    frozen_ = true;
}

size_t TargetedKmerCounter::find(uint64_t code) const {
    const size_t cap = slots_.size();
    size_t at = slot_of(code, cap);
    while (true) {
        const uint64_t key = slots_[at].key;
        if (key == code) return at;
        if (key == kEmpty) return (size_t)-1;
        if (++at == cap) at = 0;
    }
}

void TargetedKmerCounter::count_sequence(const char* s, size_t n, uint64_t& windows) {
    // Key and count of a k-mer share a 16-byte slot (one cache line per hit), and the windows are looked up a batch behind:
    // the slot of each is touched when its code is formed, so the misses of a batch overlap.
    const uint64_t mask = k_ == 32 ? ~0ull : ((1ull << (2 * k_)) - 1ull);
    constexpr size_t kBatch = 16;
    uint64_t batch[kBatch];
    size_t waiting = 0;
    const size_t cap = slots_.size();
    auto settle = [&](size_t upto) {
        for (size_t i = 0; i < upto; ++i) {
            const size_t at = find(batch[i]);
            if (at != (size_t)-1) __atomic_fetch_add(&slots_[at].count, 1ull, __ATOMIC_RELAXED);   // (workers share the table)
        }
    };
    uint64_t fwd = 0, rev = 0;
    size_t filled = 0;
    for (size_t i = 0; i < n; ++i) {
        const int b = base_code(s[i]);
        if (b < 0) { filled = 0; fwd = rev = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)b) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
        if (++filled >= k_) {
            windows += 1;
            const uint64_t code = fwd < rev ? fwd : rev;
            __builtin_prefetch(&slots_[slot_of(code, cap)]);
            batch[waiting++] = code;
            if (waiting == kBatch) { settle(kBatch); waiting = 0; }
        }
    }
    settle(waiting);
}

void TargetedKmerCounter::count(const std::string& readfile, unsigned threads) {
    if (threads == 0) threads = 1;
    freeze(threads);
    // one reader (decompression and record parsing), `threads` workers on batches of sequences; hits are relaxed atomic
    // increments on the shared table.  (Workers that parse byte ranges of the mapped file themselves — no single reader —
    // measured no better: 3.7-4.1 s against 3.1-3.3 s for 589 MB on the 256 cores of the MI355X box, 3.7-4.9 against 3.5 on 8.)
    // (a batch is the sequences of a few MB of reads back to back, a newline after each: the rolling window starts over at
    // every letter outside ACGT, so the newline is all the separation the counting needs)
    struct Batch { std::string text; };
    std::mutex mu;
    std::condition_variable cv_work, cv_room;
    std::deque<Batch> queue;
    bool done = false;
    std::exception_ptr failure;
    uint64_t windows_total = 0;
    auto worker = [&]() {
        while (true) {
            Batch b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                b = std::move(queue.front());
                queue.pop_front();
            }
            cv_room.notify_one();
            uint64_t windows = 0;
            count_sequence(b.text.data(), b.text.size(), windows);
            std::lock_guard<std::mutex> lk(mu);
            windows_total += windows;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker);
    try {
        Batch cur;
        cur.text.reserve((4u << 20) + (1u << 16));
        auto flush = [&]() {
            if (cur.text.empty()) return;
            std::unique_lock<std::mutex> lk(mu);
            cv_room.wait(lk, [&] { return queue.size() < 4u * threads; });
            queue.push_back(std::move(cur));
            cur = Batch{};
            cur.text.reserve((4u << 20) + (1u << 16));
            lk.unlock();
            cv_work.notify_one();
        };
        stream_sequences(readfile, [&](const std::string& seq) {
            cur.text.append(seq);
            cur.text.push_back('\n');
            if (cur.text.size() >= (4u << 20)) flush();
        });
        flush();
    } catch (...) {
        failure = std::current_exception();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        done = true;
    }
    cv_work.notify_all();
    for (std::thread& t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
    windows_ += windows_total;
}

size_t TargetedKmerCounter::getKmerAbundance(std::string kmer) {
    if (kmer.size() != k_) throw std::runtime_error("TargetedKmerCounter::getKmerAbundance: k-mer of length " + std::to_string(kmer.size()) + ", counter holds " + std::to_string(k_) + "-mers");
    freeze();
    uint64_t code;
    if (!encode_canonical(kmer.data(), code)) return 0;   // (letters outside ACGT: no window of a read can be this k-mer)
    const size_t at = find(code);
    if (at == (size_t)-1) {
        if (lenient_) return 0;
        throw std::runtime_error("TargetedKmerCounter::getKmerAbundance: " + kmer + " was not registered before the reads were counted");
    }
    return (size_t)slots_[at].count;
}

// ------------------------------------------------------------------ the k-mer table (behaviour: src/kmerparser.cpp)
// A row of `<prefix>_<chromosome>_kmers.tsv` is five tab-separated columns: chromosome, start, (a column this step
// does not use), the variant's unique k-mers, the flanking k-mers — the two lists comma-separated, "nan" when empty;
// rows whose first column starts with '#' are headers.  KmerRow scans a row in place: columns and list items are
// views into the line, nothing is copied until a caller asks for strings.
namespace {
struct KmerRow {
    std::string_view column[5];
    bool header = false;
    explicit KmerRow(std::string_view line) {
        size_t n = 0, at = 0;
        while (true) {
            const size_t tab = line.find('\t', at);
            if (n == 5) { n = 6; break; }  // a sixth column
            column[n++] = line.substr(at, tab == std::string_view::npos ? std::string_view::npos : tab - at);
            if (tab == std::string_view::npos) break;
            at = tab + 1;
            if (at == line.size()) break;  // (a trailing tab opens no further column: std::getline semantics of the reference's tokenizer)
        }
        if (n != 5) throw std::runtime_error("parse_kmer_line: expected 5 tab-separated fields");
        header = !column[0].empty() && column[0].front() == '#';
    }
    size_t start() const {  // leading decimal digits of column 1 (0 when there are none)
        size_t value = 0;
        std::from_chars(column[1].data(), column[1].data() + column[1].size(), value);
        return value;
    }
    template <class F>
    static void each_item(std::string_view list, F&& f) {  // comma-separated items of a list column; "nan" = no items
        if (list == "nan") return;
        while (!list.empty()) {
            const size_t comma = list.find(',');
            f(list.substr(0, comma));
            if (comma == std::string_view::npos) break;
            list.remove_prefix(comma + 1);
        }
    }
};
/** mean (integer division) of the counts inside [expected / 4, expected * 4]; `expected` itself when no count lies in
 *  the window or the window holds only zeros */
unsigned short windowed_mean(const std::vector<size_t>& counts, size_t expected) {
    const size_t lowest = expected / 4, highest = expected * 4;
    size_t sum = 0, used = 0;
    for (const size_t c : counts)
        if (c >= lowest && c <= highest) { sum += c; ++used; }
    return (unsigned short)((used && sum) ? sum / used : expected);
}
}  // namespace

void parse_kmer_line(std::string line, std::string& chrom, size_t& start, std::vector<std::string>& kmers,
                     std::vector<std::string>& flanking_kmers, bool& is_header) {
    const KmerRow row(line);
    if (row.header) { is_header = true; return; }
    chrom.assign(row.column[0]);
    start = row.start();
    KmerRow::each_item(row.column[3], [&](std::string_view k) { kmers.emplace_back(k); });
    KmerRow::each_item(row.column[4], [&](std::string_view k) { flanking_kmers.emplace_back(k); });
}

unsigned short compute_local_coverage(std::vector<std::string>& kmers, KmerCounter& read_counts, size_t kmer_coverage) {
    std::vector<size_t> counts;
    counts.reserve(kmers.size());
    for (const std::string& k : kmers) counts.push_back(read_counts.getKmerAbundance(k));
    return windowed_mean(counts, kmer_coverage);
}

// ------------------------------------------------------------------ fill_read_kmercounts
void fill_read_kmercounts(const std::string& chromosome, UniqueKmersMap* unique_kmers_map, KmerCounter& read_kmer_counts,
                          const std::string& kmers_tsv_gz, size_t kmer_coverage) {
    gzFile file = gzopen(kmers_tsv_gz.c_str(), "rb");
    if (!file) throw std::runtime_error("fill_read_kmercounts: kmer file cannot be opened.");
    auto& objects = unique_kmers_map->unique_kmers[chromosome];
    const int buffer_size = 1024;
    char buffer[buffer_size];
    std::string line;
    size_t var_index = 0;
    try {
        while (gzgets(file, buffer, buffer_size) != nullptr) {
            line += buffer;
            if (line.empty() || line.back() != '\n') continue;
            line.pop_back();
            const KmerRow row(line);
            if (!row.header) {
                if (row.column[0] != chromosome) throw std::runtime_error("fill_read_kmercounts: line of chromosome " + std::string(row.column[0]) + " in the table of " + chromosome);
                if (var_index >= objects.size()) throw std::runtime_error("fill_read_kmercounts: more lines than variants");
                UniqueKmers& u = *objects[var_index];
                if (row.start() != u.get_variant_position()) throw std::runtime_error("fill_read_kmercounts: position " + std::to_string(row.start()) + " does not match the index");
                size_t i = 0;
                KmerRow::each_item(row.column[3], [&](std::string_view k) {
                    u.update_readcount(i++, (unsigned short)read_kmer_counts.getKmerAbundance(std::string(k)));  // (size_t -> unsigned short as in the reference)
                });
                std::vector<size_t> flank_counts;
                KmerRow::each_item(row.column[4], [&](std::string_view k) { flank_counts.push_back(read_kmer_counts.getKmerAbundance(std::string(k))); });
                u.set_coverage(windowed_mean(flank_counts, kmer_coverage));
                var_index += 1;
            }
            line.clear();
        }
    } catch (...) {
        gzclose(file);
        throw;
    }
    gzclose(file);
}

void fill_read_kmercounts_all(UniqueKmersMap* unique_kmers_map, KmerCounter& read_kmer_counts, const std::string& prefix,
                              size_t kmer_coverage, unsigned threads) {
    std::vector<std::string> chromosomes;
    for (const auto& kv : unique_kmers_map->unique_kmers) chromosomes.push_back(kv.first);
    std::atomic<size_t> next{0};
    std::mutex failure_lock;
    std::exception_ptr failure;
    auto work = [&] {
        for (size_t i = next.fetch_add(1); i < chromosomes.size(); i = next.fetch_add(1)) {
            try {
                fill_read_kmercounts(chromosomes[i], unique_kmers_map, read_kmer_counts, prefix + "_" + chromosomes[i] + "_kmers.tsv.gz", kmer_coverage);
            } catch (...) {
                std::lock_guard<std::mutex> hold(failure_lock);
                if (!failure) failure = std::current_exception();
                next.store(chromosomes.size());
            }
        }
    };
    if (threads <= 1 || chromosomes.size() <= 1) work();
    else {
        // (a counter that finishes its table lazily does so here, on one thread; a strict one may refuse the k-mer itself)
        try { (void)read_kmer_counts.getKmerAbundance(std::string(unique_kmers_map->kmersize, 'A')); } catch (const std::runtime_error&) {}
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < std::min<size_t>(threads, chromosomes.size()); ++t) workers.emplace_back(work);
        for (std::thread& w : workers) w.join();
    }
    if (failure) std::rethrow_exception(failure);
}

}  // namespace pangenie
