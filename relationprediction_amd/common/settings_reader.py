"""Reader for the reference's `.exp` settings files (same format and access surface as
code/common/settings_reader.py: `read(path)` -> nested `Settings`, all leaf values are strings).

Format: a `[Section]` header opens a block; the block's lines are indented by one more TAB than the
header; `key=value` lines are leaves; blank lines are ignored; blocks nest (e.g. `[Optimizer]` ->
`[EarlyStopping]`, settings/gcn_block.exp:24-34).
"""


class Settings(object):
    """Dict-like bag of settings.  Supports `s['Key']`, `'Key' in s`, iteration over keys,
    `merge(other)` (other's keys win, code/train.py:80-86) and `put(key, value)` (train.py:76-78)."""

    def __init__(self, values=None):
        self._v = dict(values or {})

    # -- mapping surface
    def __getitem__(self, key):
        return self._v[key]

    def __contains__(self, key):
        return key in self._v

    def __iter__(self):
        return iter(self._v)

    def keys(self):
        return self._v.keys()

    def get(self, key, default=None):
        return self._v.get(key, default)

    def __repr__(self):
        return repr(self._v)

    __str__ = __repr__

    # -- mutation used by the driver
    def merge(self, other_settings):
        self._v.update(other_settings._v)

    def put(self, key, value):
        self._v[key] = value

    # -- parsing
    @staticmethod
    def _depth(line):
        n = 0
        while n < len(line) and line[n] == "\t":
            n += 1
        return n

    def parse(self, filename):
        with open(filename, "r") as f:
            self.parse_lines(f.read().splitlines())
        return self

    def parse_lines(self, lines):
        # stack[k] is the Settings object that owns lines indented by k tabs
        stack = [self]
        for raw in lines:
            if not raw.strip():
                continue
            depth = self._depth(raw)
            if depth >= len(stack):
                # deeper than any open block: the reference skips such lines too
                continue
            del stack[depth + 1:]
            text = raw.strip()
            if text.startswith("["):
                child = Settings()
                stack[depth]._v[text[1:-1]] = child
                stack.append(child)
            else:
                parts = [p.strip() for p in text.split("=")]
                stack[depth]._v[parts[0]] = parts[1]
        return self


def read(filename):
    return Settings().parse(filename)
