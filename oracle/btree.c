/*
 * oracle/btree.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 * An ordered set of u32 keys as a B-tree with the node shape of Rust's BTreeMap (B = 6: at most 11 keys per node):
 * the "reference-faithful" CPU baseline pays one tree descent per witness lookup and
 * one per insert, as acvm's WitnessMap = BTreeMap<Witness, FieldElement> does (acir/src/native_types/witness_map.rs:42-60).
 * The values stay in the dense arrays of oracle_acvm_t, so results are identical in both modes; only the cost differs.
 */
#include "pwg.h"
#include <stdlib.h>
#include <string.h>

#define BT_MAX 11 /* 2B - 1 */

typedef struct bt_node {
    uint16_t n, leaf;
    uint32_t key[BT_MAX];
    struct bt_node *child[BT_MAX + 1];
} bt_node_t;

struct oracle_btree {
    bt_node_t *root;
    size_t len;
};

static bt_node_t *node_new(int leaf) {
    bt_node_t *x = (bt_node_t *)calloc(1, sizeof *x);
    x->leaf = (uint16_t)leaf;
    return x;
}
oracle_btree_t *oracle_btree_new(void) {
    oracle_btree_t *t = (oracle_btree_t *)calloc(1, sizeof *t);
    t->root = node_new(1);
    return t;
}
static void node_free(bt_node_t *x) {
    if (!x) return;
    if (!x->leaf)
        for (int i = 0; i <= x->n; i++) node_free(x->child[i]);
    free(x);
}
void oracle_btree_free(oracle_btree_t *t) {
    if (!t) return;
    node_free(t->root);
    free(t);
}
int oracle_btree_contains(const oracle_btree_t *t, uint32_t k) {
    const bt_node_t *x = t->root;
    for (;;) {
        int i = 0;
        while (i < x->n && k > x->key[i]) i++; /* linear search inside a node, like alloc::collections::btree::search */
        if (i < x->n && k == x->key[i]) return 1;
        if (x->leaf) return 0;
        x = x->child[i];
    }
}
static void split_child(bt_node_t *x, int i) {
    bt_node_t *y = x->child[i], *z = node_new(y->leaf);
    const int mid = BT_MAX / 2; /* 5 */
    z->n = (uint16_t)(BT_MAX - mid - 1);
    memcpy(z->key, y->key + mid + 1, z->n * sizeof(uint32_t));
    if (!y->leaf) memcpy(z->child, y->child + mid + 1, (z->n + 1) * sizeof(bt_node_t *));
    y->n = (uint16_t)mid;
    memmove(x->child + i + 2, x->child + i + 1, (x->n - i) * sizeof(bt_node_t *));
    x->child[i + 1] = z;
    memmove(x->key + i + 1, x->key + i, (x->n - i) * sizeof(uint32_t));
    x->key[i] = y->key[mid];
    x->n++;
}
/* returns 1 if the key was new */
int oracle_btree_insert(oracle_btree_t *t, uint32_t k) {
    if (oracle_btree_contains(t, k)) return 0;
    bt_node_t *r = t->root;
    if (r->n == BT_MAX) {
        bt_node_t *s = node_new(0);
        s->child[0] = r;
        t->root = s;
        split_child(s, 0);
        r = s;
    }
    bt_node_t *x = r;
    for (;;) {
        int i = x->n;
        if (x->leaf) {
            while (i > 0 && k < x->key[i - 1]) { x->key[i] = x->key[i - 1]; i--; }
            x->key[i] = k;
            x->n++;
            t->len++;
            return 1;
        }
        while (i > 0 && k < x->key[i - 1]) i--;
        if (x->child[i]->n == BT_MAX) {
            split_child(x, i);
            if (k > x->key[i]) i++;
        }
        x = x->child[i];
    }
}
size_t oracle_btree_len(const oracle_btree_t *t) { return t->len; }
