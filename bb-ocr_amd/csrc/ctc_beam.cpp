// CTC beam-search decode (host side of readtext(decoder='beamsearch')).
//
// Restates easyocr/utils.py::ctcBeamSearch(mat, classes, ignore_idx, lm=None, beamWidth) as called by
// CTCLabelConverter.decode_beamsearch from easyocr/recognition.py::recognizer_predict; the reference call site
// pipeline_demo/extractor/enhanced_extractor.py:520 keeps the default decoder ('greedy'), so this is SURVEY.md §8 row f4.
// The probabilities come from the device (ctc_rows_kernel: softmax, ignore mask, renormalisation); the search itself is a
// sequential walk over T with <= beamWidth live labellings and a handful of candidate classes per step -- dictionary work
// with an ordering contract, so it stays on the host, one sequence per worker thread.
//
// Upstream's behaviour kept bit for bit (float32 arithmetic throughout, as numpy does with a float32 `mat`):
//   * a step's candidate classes are all c with mat[t][c] >= 0.5/C, INCLUDING the blank (class 0), which is then an ordinary
//     symbol of the labelling;
//   * beams are ranked by prTotal (prText == 1 without a language model) with a STABLE descending sort over the insertion
//     order of the step's dictionary;
//   * the result drops class 0 and every symbol equal to its predecessor in the labelling.
#include "kernels.h"
#include "hostpool.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <thread>
#include <vector>

namespace {

struct Beam {
    std::vector<int> lab;
    float total = 0.f, nonblank = 0.f, blank = 0.f;
};

struct BeamState {
    std::vector<Beam> entries;                    // dictionary values in insertion order
    std::map<std::vector<int>, int> index;        // labelling -> position in entries
    int at(const std::vector<int>& lab) {         // addBeam
        auto it = index.find(lab);
        if (it != index.end()) return it->second;
        const int i = (int)entries.size();
        entries.emplace_back();
        entries.back().lab = lab;
        index.emplace(lab, i);
        return i;
    }
    std::vector<int> ranked() const {             // BeamState.sort
        std::vector<int> o(entries.size());
        for (size_t i = 0; i < o.size(); ++i) o[i] = (int)i;
        std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return entries[a].total > entries[b].total; });
        return o;
    }
};

}   // namespace

void ctc_beam_search_host(const float* mat, int T, int C, int cs, int beam_width, std::vector<int>& text) {
    BeamState last;
    {
        const int i = last.at({});
        last.entries[i].blank = 1.f;
        last.entries[i].total = 1.f;
    }
    const float thr = (float)(0.5 / (double)C);
    std::vector<int> cand, lab2;
    for (int t = 0; t < T; ++t) {
        const float* p = mat + (size_t)t * cs;
        cand.clear();
        for (int c = 0; c < C; ++c)
            if (p[c] >= thr) cand.push_back(c);
        BeamState curr;
        const std::vector<int> order = last.ranked();
        const int nb = std::min<int>(beam_width, (int)order.size());
        for (int b = 0; b < nb; ++b) {
            const Beam& src = last.entries[order[b]];
            const float pr_nb = src.lab.empty() ? 0.f : src.nonblank * p[src.lab.back()];
            const float pr_b = src.total * p[0];
            {
                Beam& e = curr.entries[curr.at(src.lab)];
                e.nonblank += pr_nb;
                e.blank += pr_b;
                e.total += pr_b + pr_nb;
            }
            for (int c : cand) {
                lab2 = src.lab;
                lab2.push_back(c);
                const float ext = (!src.lab.empty() && src.lab.back() == c) ? p[c] * src.blank : p[c] * src.total;
                Beam& e = curr.entries[curr.at(lab2)];
                e.nonblank += ext;
                e.total += ext;
            }
        }
        last = std::move(curr);
    }
    text.clear();
    const std::vector<int> order = last.ranked();
    if (order.empty()) return;
    const std::vector<int>& best = last.entries[order[0]].lab;
    for (size_t i = 0; i < best.size(); ++i)
        if (best[i] != 0 && !(i > 0 && best[i - 1] == best[i])) text.push_back(best[i]);
}

void ctc_beam_search_batch(const float* probs, const int* seqs /* {first row, T} per sequence */, int nseq, int C, int cs, int beam_width,
                           std::vector<std::vector<int>>& texts, HostPool* pool) {
    texts.assign((size_t)nseq, {});
    auto one = [&](int i) { ctc_beam_search_host(probs + (size_t)seqs[2 * i] * cs, seqs[2 * i + 1], C, cs, beam_width, texts[i]); };
    if (pool) pool->parallel_for(nseq, one);
    else for (int i = 0; i < nseq; ++i) one(i);
}
