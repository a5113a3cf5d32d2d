"""CPU restatement of kreeq's candidate-error search (reference src/variants.cpp) -- TEST INFRASTRUCTURE ONLY.

Pure Python (the golden case is 23 sequences of 100 bp): each function cites the reference lines it follows.
  correct_sequences  DBG::correctSequences   src/variants.cpp:40-50
  dbg_to_variants    DBG::DBGtoVariants      :53-169   (single map range: every map is resident)
  search_variants    DBG::searchVariants     :171-310  (Dijkstra-shaped search on the Fibonacci heap of
                                                        include/fibonacci-heap.h, restated in FibHeap below)
  vcf_lines          gfalibs Report::outFile for .vcf -- the writer lives in the ABSENT gfalibs submodule; its text is
                     inferred from the one golden the reference holds, validateFiles/test.50.tst:3-37, and is
                     "parity unpinned" for anything that file does not show (several alternatives at one position).
Pinned by tests/test_oracle_golden.py::test_vcf_golden (test.50.tst, 31 records).
"""
from collections import deque

ITOC = "ACGT"
CTOI = {c: i for i, c in enumerate("ACGT")}
CTOI.update({c.lower(): i for c, i in list(CTOI.items())})
SNV, INS, DEL, COM = "SNV", "INS", "DEL", "COM"


def hash_kmer(codes, k):
    """gfalibs Kmap::hash: canonical 2-bit key, first base in the low bits; -> (key, isFw)"""
    fw = rv = 0
    for c in range(k):
        fw |= codes[c] << (2 * c)
        rv |= (3 - codes[c]) << (2 * (k - 1 - c))
    return (fw, True) if fw < rv else (rv, False)


def reverse_hash(key, k):
    return "".join(ITOC[(key >> (2 * c)) & 3] for c in range(k))


def rev_com(ch):
    return ITOC[3 - CTOI[ch]]


class _Node:
    __slots__ = ("degree", "parent", "child", "left", "right", "mark", "key", "obj")


class FibHeap:
    """include/fibonacci-heap.h, operation for operation (the search inserts every node but the source with key 0 and
    decreaseKey refuses to raise a key, :141, so the visiting order is decided by this structure alone)."""

    def __init__(self):
        self.min = None
        self.n = 0
        self.deg_table = []
        self.nodes = {}

    def size(self):
        return self.n

    def insert(self, u, key):                                            # :57-86 (the 1000-node eviction is not reached here)
        node = _Node()
        self.nodes[u] = node
        node.obj, node.key, node.degree, node.parent, node.child, node.mark = u, key, 0, None, None, False
        node.left = node.right = node
        m = self.min
        if m is not None:
            ml = m.left
            m.left = node
            node.right = m
            node.left = ml
            ml.right = node
        if m is None or m.key > node.key:
            self.min = node
        self.n += 1

    def extract_min(self):                                               # :87-126
        m = self.min
        if m is None:
            return None
        cur = m.child
        for _ in range(m.degree):
            rem = cur
            cur = cur.right
            self._existing_to_root(rem)
        self._remove_from_root(m)
        self.n -= 1
        if self.n == 0:
            self.min = None
        else:
            self.min = m.right
            ml = m.left
            self.min.left = ml
            ml.right = self.min
            self._consolidate()
        return m.obj

    def decrease_key(self, u, new_key):                                  # :127-142
        node = self.nodes[u]
        if new_key > node.key:
            return
        node.key = new_key
        if node.parent is not None and node.key < node.parent.key:
            p = node.parent
            self._cut(node)
            self._cascading_cut(p)
        if node.key < self.min.key:
            self.min = node

    def _existing_to_root(self, node):                                   # :145-164
        m = self.min
        node.parent = None
        node.mark = False
        if m is not None:
            ml = m.left
            m.left = node
            node.right = m
            node.left = ml
            ml.right = node
            if m.key > node.key:
                self.min = node
        else:
            self.min = node
            node.right = node.left = node

    def _remove_from_root(self, node):                                   # :165-179
        if node.right is not node:
            node.right.left = node.left
            node.left.right = node.right
        if node.parent is not None:
            if node.parent.degree == 1:
                node.parent.child = None
            else:
                node.parent.child = node.right
            node.parent.degree -= 1

    def _cut(self, node):
        self._remove_from_root(node)
        self._existing_to_root(node)

    def _add_child(self, parent, child):                                 # :184-201
        if parent.degree == 0:
            parent.child = child
            child.right = child.left = child
            child.parent = parent
        else:
            c1 = parent.child
            c1l = c1.left
            c1.left = child
            child.right = c1
            child.left = c1l
            c1l.right = child
        child.parent = parent
        parent.degree += 1

    def _cascading_cut(self, node):
        p = node.parent
        if p is not None:
            if not node.mark:
                node.mark = True
            else:
                self._cut(node)
                self._cascading_cut(p)

    def _link(self, high, low):
        self._remove_from_root(high)
        self._add_child(low, high)
        high.mark = False

    def _consolidate(self):                                              # :218-268
        if self.n <= 1:
            return
        self.deg_table = []
        cur = self.min
        root_cnt = 0
        it = self.min
        while True:
            root_cnt += 1
            it = it.right
            if it is self.min:
                break
        for _ in range(root_cnt):
            cons = cur
            cur = cur.right
            deg = cons.degree
            while True:
                while deg >= len(self.deg_table):
                    self.deg_table.append(None)
                if self.deg_table[deg] is None:
                    self.deg_table[deg] = cons
                    break
                other = self.deg_table[deg]
                if cons.key > other.key:
                    cons, other = other, cons
                if other is cons:
                    break
                self._link(other, cons)
                self.deg_table[deg] = None
                deg += 1
        self.min = None
        for node in self.deg_table:
            if node is not None:
                self._existing_to_root(node)


class Graph:
    """the reference's maps / maps32 as {key: (fw[4], bw[4], cov)} (an OracleDB export)"""

    def __init__(self, entries, k):
        self.k = k
        self.nodes = {int(e["key"]): ([int(x) for x in e["fw"]], [int(x) for x in e["bw"]], int(e["cov"])) for e in entries}


def build_next_kmer(key, base, fw, k):                                   # src/subgraph.cpp:581-598
    s = reverse_hash(key, k)
    s = (s + ITOC[base])[1:] if fw else (ITOC[base] + s)[:k]
    return [CTOI[c] for c in s]


def search_variants(g, source, is_source_fw, ref, targets_queue, targets_map, cache, kmer_depth, cov_cutoff=0):
    """DBG::searchVariants, src/variants.cpp:171-310.  source = key of the k-mer in front of the candidate error.
    -> (explored, [path dicts {type, sequence, refLen}])"""
    k = g.k
    destinations = []
    Q = FibHeap()
    dist = {source: 1}                                                   # :180
    prev = {}
    Q.insert(source, 1)                                                  # :181
    depth = 0
    direction = True
    explored = False
    while Q.size() > 0 and depth < kmer_depth + 1:                       # :187
        explored = False
        u = Q.extract_min()                                              # :192
        if u in prev:                                                    # :193-196
            direction = prev[u][1]
        fw_u, bw_u, _ = cache[u] if u in cache else g.nodes[u]

        def check_next(key, dirn):                                       # :197-228
            if key not in targets_map:
                if key not in cache:
                    cache[key] = g.nodes[key]                            # :204-206 (single map range: always available)
                alt = dist[u]
                if alt < 255:
                    alt += 1
                if key not in dist:
                    dist[key] = 255
                    Q.insert(key, 0)                                     # :219 (key 0: see FibHeap)
                if alt < dist[key]:
                    prev[key] = (u, dirn)
                    dist[key] = alt
                    Q.decrease_key(key, alt)                             # a no-op: alt > 0 (:141)
            return True

        edge_count = explored_count = 0
        candidates = []
        for i in range(4):                                               # :232-246
            if depth == 0:
                direction = bool(is_source_fw)
            # `direction ? fw[i] : bw[i] > covCutOff` parses as direction ? fw[i] : (bw[i] > covCutOff)   (:237)
            if (fw_u[i] != 0) if direction else (bw_u[i] > cov_cutoff):
                key, is_fw = hash_kmer(build_next_kmer(u, i, direction, k), k)
                if key != ref:                                           # :241
                    candidates.append((key, is_fw, direction))
                    edge_count += 1
        for key, is_fw, dirn in candidates:                              # :247-260
            if check_next(key, dirn if is_fw else not dirn):
                explored_count += 1
                if key in targets_map:
                    prev[key] = (u, dirn)
                    destinations.append(key)
        depth += 1
        if edge_count == explored_count or depth == kmer_depth + 1 or len(destinations) >= 10:   # :263-264
            explored = True
    paths = []
    tq = list(targets_queue)
    for destination in destinations:                                     # :266-303
        i = 0
        ref_len = (tq.index(destination) if destination in tq else len(tq)) + k
        prev_node = prev[destination][0]
        while prev_node != source:
            prev_node = prev[prev_node][0]
            i += 1
        prev_node = prev[destination][0]
        # prev[] default-constructs a missing entry as (0, false) (phmap operator[]): the source has none
        dirn = prev.get(prev_node, (0, False))[1]
        b = i - ref_len
        path = {"type": None, "sequence": [], "refLen": 0}
        if ref_len > k:
            path["type"] = COM
            path["refLen"] = ref_len - k + 1
            b = ref_len - k
        elif i == ref_len:
            path["type"] = SNV
        elif i > ref_len:
            path["type"] = DEL
            b -= 1
            prev_node = prev.get(prev_node, (0, False))[0]
            dirn = prev.get(prev_node, (0, False))[1]
        else:
            path["type"] = INS
        while b >= 0:
            s = reverse_hash(prev_node, k)
            path["sequence"].append(s[0] if dirn else rev_com(s[k - 1]))
            prev_node = prev.get(prev_node, (0, False))[0]
            dirn = prev.get(prev_node, (0, False))[1]
            b -= 1
        path["sequence"] = "".join(reversed(path["sequence"]))
        paths.append(path)
    if explored:
        for node in dist:
            cache.pop(node, None)
    return explored, paths


def dbg_to_variants(g, segment, kmer_depth, max_span, arg_order="ltr"):
    """DBG::DBGtoVariants, src/variants.cpp:53-169, for one segment (a string of ACGTacgt).
    -> [[path, ...] per position], each path with "pos" = c + k (0-based index of the first base behind k-mer c).
    arg_order: searchVariants(pair, isFw, hash(str+c+1, &isFw), ...) at :136 reads and writes isFw in one argument list;
    "ltr" passes the orientation of k-mer c (arguments evaluated left to right), "rtl" that of k-mer c + 1."""
    k = g.k
    n = len(segment)
    if n < k:
        return []
    kcount = n - k + 1
    codes = [CTOI[ch] for ch in segment]
    visited = [False] * n
    variants = []
    cache = {}
    explored_total = 0
    while explored_total < kcount:                                       # :77 (one map range: one round)
        targets_queue = deque()
        targets_map = {}
        for pos in range(max_span):                                      # :92-98
            if pos + k < kcount:
                key, _ = hash_kmer(codes[pos + k:pos + 2 * k], k)
                targets_queue.append(key)
                targets_map[key] = True
        progressed = False
        for c in range(kcount):                                          # :99-159
            if targets_queue:                                            # :104-105 (front() of an empty deque is not reached in the golden)
                targets_map.pop(targets_queue[0], None)
                targets_queue.popleft()
            if c + k + max_span < kcount:                                # :106-110
                key, _ = hash_kmer(codes[c + k + max_span:c + 2 * k + max_span], k)
                targets_map[key] = True
                targets_queue.append(key)
            if visited[c]:
                continue
            key, is_fw = hash_kmer(codes[c:c + k], k)                    # :114
            if key in g.nodes:                                           # :120-123
                if c + 1 < kcount:
                    ref, fw_next = hash_kmer(codes[c + 1:c + 1 + k], k)
                else:                                                    # :136 reads past the last k-mer; the terminator (code 4) never gives a graph key
                    ref, fw_next = -1, is_fw
                src_fw = is_fw if arg_order == "ltr" else fw_next
                ok, paths = search_variants(g, key, src_fw, ref, targets_queue, targets_map, cache, kmer_depth)
                explored_total += int(ok)
                if ok:
                    for p in paths:
                        p["pos"] = c + k                                 # :139-140
                    if paths:
                        variants.append(paths)
                    visited[c] = True
                    progressed = True
            else:                                                        # :149-152
                explored_total += 1
                visited[c] = True
                progressed = True
        if not progressed:
            break
    return variants


def segments_of(seq):
    """gfalibs splits a sequence into segments at runs of N -> [(start, text)]"""
    out, i = [], 0
    while i < len(seq):
        if seq[i] not in CTOI:
            i += 1
            continue
        j = i
        while j < len(seq) and seq[j] in CTOI:
            j += 1
        out.append((i, seq[i:j]))
        i = j
    return out


VCF_HEADER = ["##fileformat=VCFv4.2",
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
              '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">',
              "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"]


def vcf_record(header, seq, seg_start, path):
    """One VCF line of a path.  Inferred from validateFiles/test.50.tst (the writer is in the absent gfalibs):
      SNV / COM : POS = pos + 1, REF = the refLen (1 for SNV) original bases from pos, ALT = the path's bases
      INS / DEL : POS = pos (the base in front anchors the record), REF = anchor + base at pos,
                  ALT = anchor + the path's bases (+ the base at pos for DEL, which keeps it)"""
    pos = seg_start + path["pos"]
    if path["type"] in (SNV, COM):
        n = path["refLen"] if path["type"] == COM else 1
        vpos, ref, alt = pos + 1, seq[pos:pos + n], path["sequence"]
    else:
        anchor = seq[pos - 1]
        vpos, ref = pos, anchor + seq[pos]
        alt = anchor + path["sequence"] + (seq[pos] if path["type"] == DEL else "")
    return f"{header}\t{vpos}\t.\t{ref}\t{alt}\t0\tPASS\t.\tGT:GQ\t1/1:0"


def correct_sequences(g, records, kmer_depth, max_span, arg_order="ltr"):
    """DBG::correctSequences + the VCF text: records = [(header, sequence text)] in file order -> VCF lines"""
    lines = list(VCF_HEADER)
    for header, seq in records:
        for seg_start, seg in segments_of(seq):
            for paths in dbg_to_variants(g, seg, kmer_depth, max_span, arg_order):
                for p in paths:
                    lines.append(vcf_record(header, seq, seg_start, p))
    return lines
