// qd_source_hash.hip -- the hash of the sources this library was built from (build.py compiles it in; qd_source_hash() of include/qd.h).
// A unit of its own: it is the only one whose compile command changes with every source edit, so the others can be reused from
// build.py's object cache when their own inputs did not change.
#ifndef QD_SOURCE_HASH
#define QD_SOURCE_HASH ""
#endif
// the tag makes the hash findable in the file without loading it (build.py: embedded_hash)
static const char qd_source_hash_tagged[] = "QD_SOURCE_HASH=" QD_SOURCE_HASH;
extern "C" const char* qd_source_hash(void) { return qd_source_hash_tagged + sizeof("QD_SOURCE_HASH=") - 1; }
