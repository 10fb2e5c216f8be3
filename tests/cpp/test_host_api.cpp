// C++ host API smoke/parity driver (reads like the reference's src/lib.rs:152-167 test: fast == result).
// usage: test_host_api <points.bin> <scalars.bin> <expected_affine64.bin>   -> exit 0 on a bit-exact match
//        test_host_api --no-device                                           -> exit 0 if the API fails loudly without a GPU
#include <cstdio>
#include <fstream>
#include <iterator>

#include "msm_hip.hpp"

using namespace msm_webgpu;

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc == 2 && std::string(argv[1]) == "--no-device") {
    try {
      MsmContext ctx(0);
    } catch (const Error& e) {
      std::printf("expected failure: %s (code %d)\n", e.what(), e.code);
      return e.code == MSM_HIP_ERR_NO_DEVICE ? 0 : 2;
    }
    std::printf("a device is present\n");
    return 3;
  }
  if (argc != 4) return 64;
  const std::vector<uint8_t> pb = slurp(argv[1]), sb = slurp(argv[2]), want = slurp(argv[3]);
  const size_t n = sb.size() / 32;
  std::vector<G1Affine> g(n);
  std::vector<Fr> v(n);
  for (size_t i = 0; i < n; i++) {
    std::memcpy(g[i].x.data(), pb.data() + 64 * i, 32);
    std::memcpy(g[i].y.data(), pb.data() + 64 * i + 32, 32);
    std::memcpy(v[i].data(), sb.data() + 32 * i, 32);
  }
  // one-shot, as the reference calls it ...
  const G1 fast = run_webgpu_msm(g, v);
  // ... and through a persistent context
  MsmContext ctx(0);
  ctx.set_bases(g, true);
  const G1 again = ctx.msm(v);
  if (fast != again) return 1;
  // many MSMs over the resident bases: [v, 0, v] -> [result, identity, result]
  const std::vector<G1> many = ctx.msm_batch({v, std::vector<Fr>(n), v});
  if (many.size() != 3 || many[0] != fast || many[2] != fast || !many[1].to_affine().infinity) return 5;
  // the same bases with their endomorphism images (half-length scalars): same group element, compared projectively
  ctx.set_bases(g, false, MSM_HIP_BASES_ENDOMORPHISM);
  if (ctx.msm(v) != fast) return 6;
  const std::vector<G1> many2 = ctx.msm_batch({v, v});
  if (many2.size() != 2 || many2[0] != fast || many2[1] != fast) return 7;
  // several engine contexts from one process (here: three on the one GPU, pinned-buffer gather): the synchronous call and the
  // grouped asynchronous stream of jobs, plain and endomorphism bases
  {
    MultiGpuMsm node({0, 0, 0}, MSM_HIP_MGPU_GATHER_HOST);
    node.set_bases(g);
    if (node.group_size() != 2 || node.uses_rccl()) return 8;
    if (node.msm(v) != fast) return 9;
    const std::vector<std::vector<Fr>> jobs = {v, std::vector<Fr>(n), v, v, v, std::vector<Fr>(n), v, v, v};  // 9 jobs: 5 launches, three in flight
    const std::vector<G1> res = node.msm_stream(jobs);
    if (res.size() != jobs.size()) return 10;
    for (size_t k = 0; k < jobs.size(); k++)
      if ((k == 1 || k == 5) ? !res[k].to_affine().infinity : res[k] != fast) return 11;
    node.set_bases(g, MSM_HIP_BASES_ENDOMORPHISM);  // the devices then share the 8 half-length windows
    const std::vector<G1> res2 = node.msm_stream({v, v, v});
    if (res2.size() != 3 || res2[0] != fast || res2[2] != fast) return 12;
  }
  const G1Affine a = fast.to_affine();
  std::vector<uint8_t> got(64, 0);
  if (!a.infinity) {
    std::memcpy(got.data(), a.x.data(), 32);
    std::memcpy(got.data() + 32, a.y.data(), 32);
  }
  if (got != want) {
    std::printf("MISMATCH\n");
    return 1;
  }
  // error behaviour: the reference panics on infinity (lib.rs:58); here it throws
  try {
    std::vector<G1Affine> bad(1);
    bad[0].infinity = true;
    (void)points_to_bytes(bad);
    return 4;
  } catch (const std::invalid_argument&) {
  }
  std::printf("host api ok: n=%zu\n", n);
  return 0;
}
