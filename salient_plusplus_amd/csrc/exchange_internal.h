// Internal (non-ABI) interface between exchange.hip (transports) and session.hip (exchange engine).
#pragma once

#include <cstddef>

#include "spp_internal.h"

namespace spp {

// Stream-ordered point-to-point transport.  Calls are collective: every rank issues the same
// sequence of all_gather / group_begin..group_end calls.
class Transport {
 public:
  virtual ~Transport() {}
  virtual int rank() const = 0;
  virtual int world() const = 0;
  // recv[m*bytes .. (m+1)*bytes) = rank m's `send` (bytes each)
  virtual spp_status all_gather(const void* send, void* recv, size_t bytes, hipStream_t st) = 0;
  virtual spp_status group_begin() = 0;
  virtual spp_status send(const void* p, size_t bytes, int peer, hipStream_t st) = 0;
  virtual spp_status recv(void* p, size_t bytes, int peer, hipStream_t st) = 0;
  virtual spp_status group_end(hipStream_t st) = 0;
  // Give up on the communicator after a peer failed to arrive: collectives already queued on a
  // stream must leave the device (ncclCommAbort), later calls fail.  Idempotent.
  virtual void abort() = 0;
};

// seconds a rank waits for its peers inside an exchange (SPP_EXCHANGE_TIMEOUT_S, default 300)
double exchange_timeout_s();

Transport* comm_transport(spp_comm* c);

}  // namespace spp
