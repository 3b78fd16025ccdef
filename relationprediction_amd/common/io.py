"""Dictionary / triple file readers with the reference's file formats (code/common/io.py).

dictionary file: one `id<TAB>name` per line;  triple file: one `subject<TAB>relation<TAB>object`
(names) per line.  Returned triples are lists of `[subject_id, relation_id, object_id]`.
"""


def read_dictionary(filename, id_lookup=True):
    """id -> name (id_lookup=True) or name -> id (id_lookup=False); io.py:5-15."""
    table = {}
    with open(filename, "r") as f:
        for line in f:
            fields = line.strip().split("\t")
            if len(fields) < 2:
                continue
            if id_lookup:
                table[int(fields[0])] = fields[1]
            else:
                table[fields[1]] = int(fields[0])
    return table


def read_triplets(filename):
    """Yield the raw `[subject, relation, object]` name fields of each line; io.py:19-22."""
    with open(filename, "r") as f:
        for line in f:
            yield line.strip().split("\t")


def read_triplet_file(filename):
    return list(read_triplets(filename))


def read_triplets_as_list(filename, entity_dict, relation_dict):
    """Triples as integer ids, looked up through the two dictionary FILES; io.py:27-39."""
    entities = read_dictionary(entity_dict, id_lookup=False)
    relations = read_dictionary(relation_dict, id_lookup=False)
    return [[entities[s], relations[r], entities[o]] for s, r, o in read_triplets(filename)]
