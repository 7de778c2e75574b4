// Host-side helper of the programs' alignment reader (no device code): the records of an inflated BAM payload as columns, in one
// pass over the bytes -- what htslib hands the reference's io/bam.py:54-229 record by record.  Declared in include/mchap_hip.h
// (mchap_bam_count / mchap_bam_columns); the numpy construction in mchap_amd/io.py (AlignmentColumns.__init__) is the same
// table built with array operations and stays as the definition the tests hold this one against.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mchap_hip.h"

namespace {

inline int32_t le32(const uint8_t *p) {
  int32_t v;
  std::memcpy(&v, p, 4);
  return v;
}
inline uint32_t le32u(const uint8_t *p) {
  uint32_t v;
  std::memcpy(&v, p, 4);
  return v;
}
inline uint16_t le16u(const uint8_t *p) {
  uint16_t v;
  std::memcpy(&v, p, 2);
  return v;
}

// bytes of one value of an auxiliary field's type (SAM specification 4.2.4), 0 for the variable-length ones
inline int aux_size(uint8_t t) {
  switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    default: return 0;
  }
}

// the value of the record's RG:Z field (first one met walking the fields in order: what pysam's get_tag does); nullptr if none
const uint8_t *find_rg(const uint8_t *p, const uint8_t *end, int64_t *len) {
  while (p + 3 <= end) {
    const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
    p += 3;
    if (ty == 'Z' || ty == 'H') {
      const uint8_t *z = static_cast<const uint8_t *>(std::memchr(p, 0, (size_t)(end - p)));
      if (z == nullptr) return nullptr;
      if (t0 == 'R' && t1 == 'G' && ty == 'Z') {
        *len = z - p;
        return p;
      }
      p = z + 1;
    } else if (ty == 'B') {
      if (p + 5 > end) return nullptr;
      const int sz = aux_size(p[0]);
      const int64_t cnt = le32(p + 1);
      if (sz == 0 || cnt < 0) return nullptr;
      p += 5 + cnt * sz;
    } else {
      const int sz = aux_size(ty);
      if (sz == 0) return nullptr;
      p += sz;
    }
  }
  return nullptr;
}

inline uint64_t fnv1a(const uint8_t *p, int64_t n) {
  uint64_t h = 1469598103934665603ull;
  for (int64_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}

}  // namespace

extern "C" int64_t mchap_bam_count(const uint8_t *buf, int64_t n, int64_t start, int64_t *n_cigar_ops) {
  int64_t o = start, recs = 0, ops = 0;
  while (o + 4 <= n) {
    const int64_t block = le32(buf + o);
    if (block < 32 || o + 4 + block > n) break;
    ops += le16u(buf + o + 16);
    recs++;
    o += 4 + block;
  }
  if (n_cigar_ops) *n_cigar_ops = ops;
  return recs;
}

extern "C" int mchap_bam_columns(const uint8_t *buf, int64_t n, int64_t start, int64_t n_records, const char *rg_ids, int n_rg,
                                 int64_t *offset, int32_t *ref_id, int32_t *pos, int64_t *end, int32_t *mapq, int32_t *flag,
                                 int64_t *seq_off, int64_t *qual_off, int64_t *rg, int64_t *qname_id, int64_t *seg_first,
                                 int64_t *c_rec, int64_t *c_op, int64_t *c_len, int64_t *c_ref0, int64_t *c_read0) {
  // the read group ids of the header: NUL-separated
  std::vector<const char *> ids((size_t)n_rg);
  std::vector<size_t> id_len((size_t)n_rg);
  {
    const char *p = rg_ids;
    for (int g = 0; g < n_rg; g++) {
      ids[(size_t)g] = p;
      id_len[(size_t)g] = std::strlen(p);
      p += id_len[(size_t)g] + 1;
    }
  }
  // query names -> ids (the record that first carries the name): open addressing, names compared byte for byte
  size_t cap = 16;
  while (cap < (size_t)(2 * n_records + 1)) cap <<= 1;
  std::vector<int64_t> table(cap, -1);
  int64_t o = start, seg = 0;
  for (int64_t r = 0; r < n_records; r++) {
    if (o + 4 > n) return MCHAP_ERR_BAD_ARG;
    const int64_t block = le32(buf + o);
    if (block < 32 || o + 4 + block > n) return MCHAP_ERR_BAD_ARG;
    const uint8_t *c = buf + o + 4;
    const int64_t l_name = c[8], n_cig = le16u(c + 12), l_seq = le32(c + 16);
    offset[r] = o;
    ref_id[r] = le32(c);
    pos[r] = le32(c + 4);
    mapq[r] = c[9];
    flag[r] = le16u(c + 14);
    const int64_t name_off = o + 36, cig_off = name_off + l_name;
    seq_off[r] = cig_off + 4 * n_cig;
    qual_off[r] = seq_off[r] + (l_seq + 1) / 2;
    const int64_t tag_off = qual_off[r] + l_seq, rec_end = o + 4 + block;
    if (l_seq < 0 || tag_off > rec_end) return MCHAP_ERR_BAD_ARG;
    // query name (without its NUL)
    {
      const uint8_t *nm = buf + name_off;
      const int64_t ln = l_name > 0 ? l_name - 1 : 0;
      size_t slot = (size_t)fnv1a(nm, ln) & (cap - 1);
      for (;;) {
        const int64_t other = table[slot];
        if (other < 0) {
          table[slot] = r;
          qname_id[r] = r;
          break;
        }
        const int64_t o2 = offset[other];
        const int64_t ln2 = buf[o2 + 12] > 0 ? buf[o2 + 12] - 1 : 0;
        if (ln2 == ln && std::memcmp(buf + o2 + 36, nm, (size_t)ln) == 0) {
          qname_id[r] = other;
          break;
        }
        slot = (slot + 1) & (cap - 1);
      }
    }
    // read group
    {
      int64_t len = 0;
      const uint8_t *v = find_rg(buf + tag_off, buf + rec_end, &len);
      int64_t gi = -1;
      if (v != nullptr)
        for (int g = 0; g < n_rg; g++)
          if ((int64_t)id_len[(size_t)g] == len && std::memcmp(ids[(size_t)g], v, (size_t)len) == 0) {
            gi = g;
            break;
          }
      rg[r] = gi;
    }
    // CIGAR operations with the reference / read offset at which each starts
    seg_first[r] = seg;
    int64_t ref_at = pos[r], read_at = 0;
    for (int64_t k = 0; k < n_cig; k++) {
      const uint32_t w = le32u(buf + cig_off + 4 * k);
      const int64_t op = w & 15, ln = w >> 4;
      c_rec[seg] = r;
      c_op[seg] = op;
      c_len[seg] = ln;
      c_ref0[seg] = ref_at;
      c_read0[seg] = read_at;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_at += ln;   // M D N = X consume the reference
      if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) read_at += ln;  // M I S = X consume the read
      seg++;
    }
    end[r] = ref_at;
    o = rec_end;
  }
  seg_first[n_records] = seg;
  return MCHAP_OK;
}
