"""Negative sampling (reference: code/common/auxilliaries.py).

`NegativeSampler.transform` (:13-33): the batch is tiled (rate + 1) times; the first copy keeps label 1,
every further row gets label 0 and either its object (fair coin = 1) or its subject replaced by a uniformly
drawn entity.  Row (i, j) of the reference's double loop reads `choices[i + j*size]`, `values[i + j*size]`
and writes row `size + i + j*size` — one draw per negative row, in row order — so the vectorised form below
is the same function of the same random streams (`np.random.binomial` then `np.random.randint`).
Corrupted triples are NOT checked against known positives (that is `transform_exclusive`, which no
BASELINE setting calls)."""
import numpy as np


class NegativeSampler(object):
    negative_sample_rate = None
    n_entities = None

    def __init__(self, negative_sample_rate, n_entities):
        self.negative_sample_rate = int(negative_sample_rate)
        self.n_entities = int(n_entities)

    def transform(self, triplets, rng=None):
        """`rng`: a numpy RandomState to draw from instead of the global one (background batch producers)."""
        rng = np.random if rng is None else rng
        triplets = np.asarray(triplets)
        size_of_batch = len(triplets)
        number_to_generate = size_of_batch * self.negative_sample_rate
        new_labels = np.zeros(size_of_batch * (self.negative_sample_rate + 1), dtype=np.float32)
        new_indexes = np.tile(triplets, (self.negative_sample_rate + 1, 1)).astype(np.int32)
        new_labels[:size_of_batch] = 1
        choices = rng.binomial(1, 0.5, number_to_generate).astype(bool)
        values = rng.randint(self.n_entities, size=number_to_generate)
        negatives = new_indexes[size_of_batch:]
        negatives[choices, 2] = values[choices]
        negatives[~choices, 0] = values[~choices]
        return new_indexes, new_labels

    def set_known_positives(self, triplets):
        self.objs, self.subs = {}, {}
        for s, r, o in np.asarray(triplets):
            self.objs.setdefault(s, set()).add((r, o))
            self.subs.setdefault(o, set()).add((r, s))

    def transform_exclusive(self, triplets):
        """As `transform`, redrawing every corruption that is a known positive (:48-70)."""
        triplets = np.asarray(triplets)
        size_of_batch = len(triplets)
        number_to_generate = size_of_batch * self.negative_sample_rate
        new_labels = np.zeros(size_of_batch * (self.negative_sample_rate + 1), dtype=np.float32)
        new_indexes = np.tile(triplets, (self.negative_sample_rate + 1, 1)).astype(np.int32)
        new_labels[:size_of_batch] = 1
        choices = np.random.binomial(1, 0.5, number_to_generate)
        for k in range(number_to_generate):
            row = new_indexes[size_of_batch + k]
            if choices[k]:
                row[2] = np.random.randint(self.n_entities)
                while (row[1], row[2]) in self.objs.get(row[0], ()):
                    row[2] = np.random.randint(self.n_entities)
            else:
                row[0] = np.random.randint(self.n_entities)
                while (row[1], row[0]) in self.subs.get(row[2], ()):
                    row[0] = np.random.randint(self.n_entities)
        return new_indexes, new_labels
