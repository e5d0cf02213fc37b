// N3: candidate matchers on device — the proposals the evaluation sampler and the submission flow consume.
//   mi_match_common_items_i32   UsersWithCommonItemsMatcher.get_matches (data/matching/users_with_common_purchases.py:14-26)
//   mi_match_same_location_i32  UsersSameLocationMatcher.get_matches (data/matching/fashion/users_same_location.py:15-25)
// The reference materialises, per user, every article list of every user who bought any of the user's articles and
// then keeps the first k; here one wavefront per query user walks that concatenation in the same order — the user's
// articles in list order, each article's users in list order (the user itself included), each such user's articles in
// list order — and stops at k entries, copying each list with all its lanes.  Integer work: bit-exact.
#include "common.hpp"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void match_common_items_kernel(int64_t n_q, const int64_t* __restrict__ q_users,
                                                                    const int32_t* __restrict__ uptr,
                                                                    const int32_t* __restrict__ uidx,
                                                                    const int32_t* __restrict__ aptr,
                                                                    const int32_t* __restrict__ aidx, int32_t k,
                                                                    int32_t* __restrict__ out /* [n_q, k], -1 padded */,
                                                                    int32_t* __restrict__ out_n) {
    const int64_t q = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / MI_WAVE;
    if (q >= n_q) return;
    const int lane = mi_lane();
    const int64_t u = q_users ? q_users[q] : q;
    int32_t* dst = out + q * (int64_t)k;
    int have = 0;  // wave-uniform
    for (int32_t pa = uptr[u]; pa < uptr[u + 1] && have < k; ++pa) {
        const int32_t a = uidx[pa];
        for (int32_t pv = aptr[a]; pv < aptr[a + 1] && have < k; ++pv) {
            const int32_t v = aidx[pv];
            const int32_t b = uptr[v];
            const int take = min(uptr[v + 1] - b, k - have);
            for (int j = lane; j < take; j += MI_WAVE) dst[have + j] = uidx[b + j];
            have += take;
        }
    }
    for (int j = have + lane; j < k; j += MI_WAVE) dst[j] = -1;
    if (out_n && lane == 0) out_n[q] = have;
}

// UsersSameLocationMatcher.get_matches (data/matching/fashion/users_same_location.py:15-25): the article lists of
// every customer at the query user's location (customers_per_location order, the user itself included), concatenated,
// first k.  Same walk as above with the location's customer list in place of the co-purchasers.
__global__ __launch_bounds__(kBlock) void match_same_location_kernel(int64_t n_q, const int64_t* __restrict__ q_users,
                                                                     const int32_t* __restrict__ loc_of_user,
                                                                     const int32_t* __restrict__ lptr,
                                                                     const int32_t* __restrict__ lidx,
                                                                     const int32_t* __restrict__ uptr,
                                                                     const int32_t* __restrict__ uidx, int32_t k,
                                                                     int32_t* __restrict__ out, int32_t* __restrict__ out_n) {
    const int64_t q = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / MI_WAVE;
    if (q >= n_q) return;
    const int lane = mi_lane();
    const int64_t u = q_users ? q_users[q] : q;
    int32_t* dst = out + q * (int64_t)k;
    int have = 0;  // wave-uniform
    const int32_t loc = loc_of_user[u];
    if (loc >= 0) {
        for (int32_t pv = lptr[loc]; pv < lptr[loc + 1] && have < k; ++pv) {
            const int32_t v = lidx[pv];
            const int32_t b = uptr[v];
            const int take = min(uptr[v + 1] - b, k - have);
            for (int j = lane; j < take; j += MI_WAVE) dst[have + j] = uidx[b + j];
            have += take;
        }
    }
    for (int j = have + lane; j < k; j += MI_WAVE) dst[j] = -1;
    if (out_n && lane == 0) out_n[q] = have;
}

}  // namespace

extern "C" int mi_match_same_location_i32(int64_t n_queries, const int64_t* query_users, const int32_t* location_of_user,
                                          const int32_t* loc_ptr, const int32_t* loc_idx, const int32_t* users_ptr,
                                          const int32_t* users_idx, int32_t k, int32_t* out, int32_t* out_count,
                                          mi_stream_t stream) {
    MI_CHECK_ARG(n_queries >= 0 && k > 0);
    if (n_queries == 0) return 0;
    MI_CHECK_ARG(location_of_user && loc_ptr && loc_idx && users_ptr && users_idx && out);
    hipLaunchKernelGGL(match_same_location_kernel, dim3((unsigned)mi_ceil_div(n_queries * MI_WAVE, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n_queries, query_users, location_of_user, loc_ptr, loc_idx, users_ptr, users_idx, k,
                       out, out_count);
    return mi_launch_status();
}

extern "C" int mi_match_common_items_i32(int64_t n_queries, const int64_t* query_users, const int32_t* users_ptr,
                                         const int32_t* users_idx, const int32_t* articles_ptr,
                                         const int32_t* articles_idx, int32_t k, int32_t* out, int32_t* out_count,
                                         mi_stream_t stream) {
    MI_CHECK_ARG(n_queries >= 0 && k > 0);
    if (n_queries == 0) return 0;
    MI_CHECK_ARG(users_ptr && users_idx && articles_ptr && articles_idx && out);
    hipLaunchKernelGGL(match_common_items_kernel, dim3((unsigned)mi_ceil_div(n_queries * MI_WAVE, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n_queries, query_users, users_ptr, users_idx, articles_ptr, articles_idx, k, out,
                       out_count);
    return mi_launch_status();
}
