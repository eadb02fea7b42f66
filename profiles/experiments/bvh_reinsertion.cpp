// Experiment, not compiled into the library: host-side reinsertion pass over the binary tree of bvh_build.hip
// (it sat between the build rounds and k_emit_nodes, switched on by MIPT_BVH_REINSERT=<passes>).  Result: see
// profiles/r02_tree_experiments.md.

// ---------------------------------------------------------------------------------------
// Tree optimisation by reinsertion (after Bittner, Hapala, Havran: "Fast insertion-based optimization
// of bounding volume hierarchies", 2013), on the host, between the build and the traversal layout.
// A pass takes the nodes in descending order of surface area; each is cut out of the tree (its
// parent goes, its sibling moves up) and put back where it adds the least surface area, found by a
// branch-and-bound search from the root.  Leaves keep their triangle ranges, so the triangle order
// is untouched.  The result is written back level by level with siblings adjacent, as the build
// rounds produce it and the collapse to wide nodes expects it.
#include <queue>
#include <algorithm>
namespace {
struct ONode { float mn[3], mx[3]; int parent, left, right; int src; };   // src: index of the leaf's BNode, -1 for inner nodes

inline float o_area(const float* mn, const float* mx) { float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return dx * dy + dy * dz + dz * dx; }
inline float o_union_area(const ONode& a, const ONode& b)
{
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = std::min(a.mn[k], b.mn[k]); mx[k] = std::max(a.mx[k], b.mx[k]); }
    return o_area(mn, mx);
}
// recomputes the boxes from node i upwards; stops at the first ancestor whose box does not change
// (the box of i itself may be stale, e.g. a parent node reused in a new place: its ancestors are always visited)
inline void o_refit_up(std::vector<ONode>& t, int i, bool staleFirst = false)
{
    while (i >= 0) {
        ONode& n = t[i];
        const ONode &l = t[n.left], &r = t[n.right];
        bool same = true;
        for (int k = 0; k < 3; k++) {
            float a = std::min(l.mn[k], r.mn[k]), b = std::max(l.mx[k], r.mx[k]);
            if (a != n.mn[k] || b != n.mx[k]) same = false;
            n.mn[k] = a; n.mx[k] = b;
        }
        if (same && !staleFirst) break;
        staleFirst = false;
        i = n.parent;
    }
}
}

static inline float bn_key2f(int32_t i) { int32_t j = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &j, 4); return f; }
static inline int32_t bn_f2key(float f) { int32_t i; memcpy(&i, &f, 4); return i >= 0 ? i : i ^ 0x7fffffff; }

// nodes: the binary tree as built (root 0, children left and left + 1).  Returns the optimised tree in
// the same form in `nodes` and its level ends in levelEnd; sahBefore / sahAfter = sum of inner areas + leaf areas * count over the root area.
static void optimize_tree_host(std::vector<BNode>& nodes, std::vector<int32_t>& levelEnd, int passes, float batchShare, double* sahBefore, double* sahAfter)
{
    const int n = (int)nodes.size();
    std::vector<ONode> t(n);
    for (int i = 0; i < n; i++) {
        const BNode& b = nodes[i];
        for (int k = 0; k < 3; k++) { t[i].mn[k] = bn_key2f(b.bminI[k]); t[i].mx[k] = bn_key2f(b.bmaxI[k]); }
        t[i].parent = -1;
        if (b.state == NODE_INNER) { t[i].left = b.left; t[i].right = b.left + 1; t[i].src = -1; }
        else { t[i].left = t[i].right = -1; t[i].src = i; }
    }
    for (int i = 0; i < n; i++) if (t[i].left >= 0) { t[t[i].left].parent = i; t[t[i].right].parent = i; }
    auto sah = [&]() {
        double s = 0;
        for (int i = 0; i < n; i++) { double a = o_area(t[i].mn, t[i].mx); s += t[i].left >= 0 ? a : a * nodes[t[i].src].count; }
        return s / std::max((double)o_area(t[0].mn, t[0].mx), 1e-300);
    };
    // (nodes cut out of the tree during the loop are put back before the next one is taken: the array always holds one tree)
    if (sahBefore) *sahBefore = sah();
    std::vector<int> order;
    struct QE { float induced; int node; bool operator<(const QE& o) const { return induced > o.induced; } };
    for (int pass = 0; pass < passes; pass++) {
        order.clear();
        for (int i = 1; i < n; i++) if (t[i].parent > 0) order.push_back(i);      // not the root, not its children
        std::sort(order.begin(), order.end(), [&](int a, int b) { return o_area(t[a].mn, t[a].mx) > o_area(t[b].mn, t[b].mx); });
        size_t take = (size_t)(order.size() * batchShare);
        if (take < 1) take = order.size();
        for (size_t oi = 0; oi < take && oi < order.size(); oi++) {
            const int L = order[oi];
            const int P = t[L].parent;
            if (P <= 0) continue;                                   // became a child of the root meanwhile
            const int G = t[P].parent;
            const int S = (t[P].left == L) ? t[P].right : t[P].left;
            // cut L and P out: S takes P's place under G
            if (t[G].left == P) t[G].left = S; else t[G].right = S;
            t[S].parent = G;
            o_refit_up(t, G);
            // best place for L: the node X that minimises area(X u L) + the growth of X's ancestors
            const float areaL = o_area(t[L].mn, t[L].mx);
            float bestCost = FLT_MAX; int best = S;
            std::priority_queue<QE> pq;
            pq.push({ 0.0f, 0 });
            while (!pq.empty()) {
                QE e = pq.top(); pq.pop();
                if (e.induced + areaL >= bestCost) break;
                const ONode& x = t[e.node];
                float direct = o_union_area(x, t[L]);
                float total = e.induced + direct;
                if (total < bestCost) { bestCost = total; best = e.node; }
                float childInduced = total - o_area(x.mn, x.mx);
                if (x.left >= 0 && childInduced + areaL < bestCost) { pq.push({ childInduced, x.left }); pq.push({ childInduced, x.right }); }
            }
            // P becomes the parent of best and L, in best's place
            const int X = best, XP = t[X].parent;
            if (XP < 0) {
                // above the root: keep index 0 as the root by moving the old root's content into P
                t[P] = t[0];
                t[t[P].left].parent = P; t[t[P].right].parent = P;
                t[0].left = P; t[0].right = L; t[0].parent = -1; t[0].src = -1;
                t[P].parent = 0; t[L].parent = 0;
                o_refit_up(t, P, true);
            } else {
                if (t[XP].left == X) t[XP].left = P; else t[XP].right = P;
                t[P].parent = XP; t[P].left = X; t[P].right = L; t[P].src = -1;
                t[X].parent = P; t[L].parent = P;
                o_refit_up(t, P, true);
            }
        }
    }
    if (sahAfter) *sahAfter = sah();
    // write back level by level, siblings adjacent
    std::vector<BNode> out(n);
    std::vector<int> cur{ 0 }, nxt, newId(n, -1);
    levelEnd.clear();
    int used = 1; newId[0] = 0;
    int depth = 0;
    std::vector<int> depthOf(n, 0);
    while (!cur.empty()) {
        nxt.clear();
        for (int i : cur) {
            if (t[i].left >= 0) { newId[t[i].left] = used++; newId[t[i].right] = used++; nxt.push_back(t[i].left); nxt.push_back(t[i].right); }
            depthOf[i] = depth;
        }
        levelEnd.push_back(used - (int)nxt.size());     // nodes of depth <= this level
        cur.swap(nxt); depth++;
    }
    for (int i = 0; i < n; i++) {
        BNode b = {};
        if (t[i].left >= 0) { b.state = NODE_INNER; b.left = newId[t[i].left]; b.count = 0; b.start = 0; b.axis = 0; }
        else { b = nodes[t[i].src]; b.left = -1; }
        for (int k = 0; k < 3; k++) { b.bminI[k] = bn_f2key(t[i].mn[k]); b.bmaxI[k] = bn_f2key(t[i].mx[k]); }
        b.depth = depthOf[i];
        out[newId[i]] = b;
    }
    nodes.swap(out);
}


// ---- the call site inside mipt_build_bvh ----
#if 0
    if (const char* e = getenv("MIPT_BVH_REINSERT")) {
        int passes = atoi(e);
        if (passes > 0 && h.nodeCount > 3) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<BNode> hn((size_t)h.nodeCount);
            HIPCHK(hipMemcpy(hn.data(), nodes, sizeof(BNode) * (size_t)h.nodeCount, hipMemcpyDeviceToHost));
            double s0 = 0, s1 = 0;
            const char* sh = getenv("MIPT_BVH_REINSERT_SHARE");
            optimize_tree_host(hn, levelEnd, passes, sh ? (float)atof(sh) : 1.0f, &s0, &s1);
            // posNode maps a triangle position to its leaf: leaves moved
            std::vector<int32_t> pn((size_t)n);
            for (int i = 0; i < h.nodeCount; i++) if (hn[i].state != NODE_INNER) for (int p = hn[i].start; p < hn[i].start + hn[i].count; p++) pn[p] = i;
            HIPCHK(hipMemcpy(nodes, hn.data(), sizeof(BNode) * (size_t)h.nodeCount, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(posNode[cur], pn.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
            int md = 0;
            for (int i = 0; i < h.nodeCount; i++) if (hn[i].state != NODE_INNER && hn[i].depth > md) md = hn[i].depth;
            h.maxDepth = md;
            HIPCHK(hipMemcpy(&ctr->maxDepth, &md, sizeof(int32_t), hipMemcpyHostToDevice));
            if (getenv("MIPT_BUILD_VERBOSE"))
                fprintf(stderr, "bvh build: reinsertion, %d passes: SAH %.3f -> %.3f, depth %d, %.1f ms on the host\n", passes, s0, s1, md,
                        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    }
#endif
