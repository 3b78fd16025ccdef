"""Link-prediction evaluation (reference: code/common/evaluation.py).

`Scorer` keeps the reference's surface (register_data / register_degrees / register_model /
finalize_frequency_computation / compute_scores(...).get_summary() with `results['Raw'|'Filtered']['MRR'|'H@n']`)
and its definitions (MrrScore.append_line, :148-153: raw rank = #{score >= score[gold]}, filtered rank =
raw - #{known with score >= score[gold]} + 1; ranks of the subject side first, then the object side, per
chunk of 1000 triples, :334-389).  What changes is where the work happens: the reference encodes the full graph
and materialises a [1000, V] score matrix in numpy for every chunk and side; here the graph is encoded once per
`compute_scores` and the ranks are counted on the device (rgcn_rank_device, csrc/ranking.hip).

Metric=Accuracy (the *_accuracy.txt datasets: (positive, negative) pairs on consecutive lines; no BASELINE config uses
it) is the share of pairs whose positive outscores its negative, from one `model.score` of the listed triples (:178-209,
:311-326)."""
import math

import numpy as np


class MrrSummary(object):
    calculate_hits_at = [1, 3, 10]

    def __init__(self, raw_ranks, filtered_ranks, in_degrees=None, out_degrees=None, vertex_freqs=None,
                 relation_freqs=None):
        self.results = {'Raw': {}, 'Filtered': {}}
        for name, ranks in (('Raw', raw_ranks), ('Filtered', filtered_ranks)):
            ranks = np.asarray(ranks, dtype=np.float64)
            self.results[name][self.mrr_string()] = self.get_mrr(ranks)
            for h in self.calculate_hits_at:
                self.results[name][self.hits_string(h)] = self.get_hits_at_n(ranks, h)
            if vertex_freqs is not None:
                self.results[name][self.freq_string()] = list(zip(1.0 / ranks, vertex_freqs, relation_freqs))
            if in_degrees is not None:
                self.results[name][self.degree_string()] = self.get_individual_degree_scores(ranks, in_degrees,
                                                                                             out_degrees)

    @staticmethod
    def get_individual_degree_scores(ranks, in_degrees, out_degrees):
        """What results['Degree'] holds in the reference (:16-17,28-36): per evaluation, (in-degree of the fixed
        entity, 1/rank) and (out-degree, 1/rank) -- two lists in evaluation order, not bucket means."""
        reciprocal = 1.0 / np.asarray(ranks, dtype=np.float64)
        return ([(int(deg), float(rr)) for deg, rr in zip(in_degrees, reciprocal)],
                [(int(deg), float(rr)) for deg, rr in zip(out_degrees, reciprocal)])

    @staticmethod
    def get_degree_scores(ranks, in_degrees, out_degrees):
        """Mean reciprocal rank bucketed by degree (:38-62; not stored in `results` by the reference either).  Bucket
        label i stands for degree i + 1, as in the reference's lists."""
        res = []
        reciprocal = 1.0 / np.asarray(ranks, dtype=np.float64)
        for degrees in (in_degrees, out_degrees):
            degrees = np.asarray(degrees, dtype=np.int64)
            if (degrees < 1).any():
                raise ValueError("degree buckets need degrees >= 1 (the reference indexes bucket degree - 1)")
            sums = np.bincount(degrees - 1, weights=reciprocal)
            counts = np.bincount(degrees - 1)
            res.append([(int(i), float(sums[i] / counts[i])) for i in np.flatnonzero(counts)])
        return tuple(res)

    def dump_degrees(self, in_filename, out_filename, filter='Filtered'):
        """one line per evaluation: label + 1, reciprocal rank (:99-111)"""
        for filename, pairs in zip((in_filename, out_filename), self.results[filter][self.degree_string()]):
            with open(filename, 'w+') as f:
                f.writelines('%s\t%s\n' % (label + 1, value) for label, value in pairs)

    def dump_frequencies(self, vertex_filename, relation_filename, filter='Filtered'):
        """one line per evaluation: reciprocal rank with the entity's mean relation frequency / the relation's (:117-128)"""
        rows = self.results[filter][self.freq_string()]
        with open(vertex_filename, 'w+') as vf, open(relation_filename, 'w+') as rf:
            for reciprocal, vertex_freq, relation_freq in rows:
                vf.write('%s\t%s\n' % (reciprocal, vertex_freq))
                rf.write('%s\t%s\n' % (reciprocal, relation_freq))

    def mrr_string(self):
        return 'MRR'

    def hits_string(self, n):
        return 'H@' + str(n)

    def degree_string(self):
        return "Degree"

    def freq_string(self):
        return "Frequency"

    def pretty_print(self):
        print('\tRaw\tFiltered')
        for item in [self.mrr_string()] + [self.hits_string(h) for h in self.calculate_hits_at]:
            print(item, end='\t')
            print(str(round(self.results['Raw'][item], 3)), end='\t')
            print(str(round(self.results['Filtered'][item], 3)))

    def get_mrr(self, ranks):
        return float(np.mean(1.0 / np.asarray(ranks, dtype=np.float64))) if len(ranks) else 0.0

    def get_hits_at_n(self, ranks, n):
        return float(np.mean(np.asarray(ranks) <= n)) if len(ranks) else 0.0


class MrrScore(object):
    def __init__(self, dataset):
        n = len(dataset) * 2
        self.raw_ranks = np.zeros(n, dtype=np.int64)
        self.filtered_ranks = np.zeros(n, dtype=np.int64)
        self.in_degree = np.zeros(n, dtype=np.int64)
        self.out_degree = np.zeros(n, dtype=np.int64)
        self.vertex_freq = np.zeros(n, dtype=np.float64)
        self.relation_freq = np.zeros(n, dtype=np.float64)
        self.pointer = 0

    def append_lines(self, raw, filtered, in_degree, out_degree, vertex_freq, relation_freq):
        k = len(raw)
        sl = slice(self.pointer, self.pointer + k)
        self.raw_ranks[sl], self.filtered_ranks[sl] = raw, filtered
        self.in_degree[sl], self.out_degree[sl] = in_degree, out_degree
        self.vertex_freq[sl], self.relation_freq[sl] = vertex_freq, relation_freq
        self.pointer += k

    def get_summary(self):
        n = self.pointer
        return MrrSummary(self.raw_ranks[:n], self.filtered_ranks[:n], self.in_degree[:n], self.out_degree[:n],
                          self.vertex_freq[:n], self.relation_freq[:n])

    def summarize(self):
        self.get_summary().pretty_print()

    def print_to_file(self, filename):
        with open(filename, 'w+') as outfile:
            for raw, filtered in zip(self.raw_ranks[:self.pointer], self.filtered_ranks[:self.pointer]):
                print(str(raw) + '\t' + str(filtered), file=outfile)


class AccuracySummary(object):
    """code/common/evaluation.py:178-196 (`results` is per summary here; the reference keeps ONE class-level dict)"""

    def __init__(self, predictions):
        self.results = {'Filtered': {}, 'Raw': {}}
        self.results['Filtered'][self.accuracy_string()] = np.mean(predictions)

    def dump_degrees(self, in_file, out_file):
        pass

    def accuracy_string(self):
        return 'Accuracy'

    def pretty_print(self):
        for item in [self.accuracy_string()]:
            print(item, end='\t')
            print(str(round(self.results['Filtered'][item], 3)), end='\n')


class AccuracyScore(object):
    """code/common/evaluation.py:199-209"""

    def append_all(self, evaluations):
        self.predictions = evaluations

    def summarize(self):
        self.get_summary().pretty_print()

    def get_summary(self):
        return AccuracySummary(self.predictions)


class Scorer(object):
    chunk_size = 1000

    def __init__(self, settings):
        self.known_object_triples = {}
        self.known_subject_triples = {}
        self.in_degree = {}
        self.out_degree = {}
        self.relation_freqs = {}
        self.avg_freq = {}
        self.settings = settings
        self.model = None

    @staticmethod
    def extend_triple_dict(dictionary, triplets, object_list=True):
        for s, r, o in np.asarray(triplets):
            key, value = ((s, r), o) if object_list else ((o, r), s)
            lst = dictionary.setdefault((int(key[0]), int(key[1])), [])
            if value not in lst:
                lst.append(int(value))

    def register_data(self, triples):
        for s, r, o in np.asarray(triples):
            for v in (int(s), int(o)):
                self.in_degree.setdefault(v, 0)
                self.out_degree.setdefault(v, 0)
            self.relation_freqs[int(r)] = self.relation_freqs.get(int(r), 0) + 1
        self.extend_triple_dict(self.known_subject_triples, triples, object_list=False)
        self.extend_triple_dict(self.known_object_triples, triples)

    def register_degrees(self, triples):
        for s, _, o in np.asarray(triples):
            self.in_degree[int(o)] += 1
            self.out_degree[int(s)] += 1

    def register_model(self, model):
        self.model = model

    def finalize_frequency_computation(self, triples):
        counts = {}
        for s, r, o in np.asarray(triples):
            for v in (int(s), int(o)):
                self.avg_freq[v] = self.avg_freq.get(v, 0) + self.relation_freqs[int(r)]
                counts[v] = counts.get(v, 0) + 1
        for k in counts:
            self.avg_freq[k] /= float(counts[k])

    def get_degrees(self, vertex):
        return self.in_degree[vertex], self.out_degree[vertex]

    def compute_scores(self, triples, verbose=False):
        if self.settings['Metric'] == 'MRR':
            return self.compute_mrr_scores(triples, verbose=verbose)
        if self.settings['Metric'] == 'Accuracy':
            return self.compute_accuracy_scores(triples, verbose=verbose)
        raise NotImplementedError("Evaluation.Metric=%s (MRR and Accuracy are what the reference has)"
                                  % self.settings['Metric'])

    def compute_accuracy_scores(self, triples, verbose=False):
        """code/common/evaluation.py:311-326: rows 2k / 2k + 1 are a (positive, negative) pair; a tie counts as wrong"""
        score = AccuracyScore()
        if verbose:
            print("Evaluating accuracies...")
        score_vector = np.asarray(self.model.score(triples))
        score.append_all(score_vector[::2] > score_vector[1::2])
        return score

    def compute_mrr_scores(self, triples, verbose=False):
        triples = np.asarray(triples)
        score = MrrScore(triples)
        n_chunks = math.ceil(len(triples) / self.chunk_size)
        for chunk in range(n_chunks):
            self.evaluate_mrr(score, triples[chunk * self.chunk_size:(chunk + 1) * self.chunk_size], verbose)
        return score

    def _filter_csr(self, triples, subject_side):
        ptr, idx = np.zeros(len(triples) + 1, dtype=np.int64), []
        for i, (s, r, o) in enumerate(triples):
            known = (self.known_subject_triples[(int(o), int(r))] if subject_side
                     else self.known_object_triples[(int(s), int(r))])
            idx.extend(known)
            ptr[i + 1] = len(idx)
        return ptr, np.asarray(idx, dtype=np.int32)

    def evaluate_mrr(self, score, triples, verbose):
        """Subject side of the chunk, then its object side (the reference's row order)."""
        graph = self.model.test_graph
        for subject_side in (True, False):
            if verbose:
                print("Evaluating %s..." % ("subjects" if subject_side else "objects"))
            ptr, idx = self._filter_csr(triples, subject_side)
            raw, filtered = self.model.device_ranks(graph, triples, not subject_side, ptr, idx)
            fixed = triples[:, 2] if subject_side else triples[:, 0]
            score.append_lines(raw, filtered,
                               [self.in_degree[int(v)] for v in fixed], [self.out_degree[int(v)] for v in fixed],
                               [self.avg_freq.get(int(v), 0.0) for v in fixed],
                               [self.relation_freqs[int(r)] for r in triples[:, 1]])
