// Shared definitions of the opaque C-ABI handles and the exception -> status translation.
#pragma once
#include "../../include/mgamd.h"
#include "../../include/mgamd_dev.h"
#include "level_tables.hpp"
#include "transfer_tables.hpp"
#include "partition.hpp"
#include "amg.hpp"

#include <memory>
#include <string>

namespace mgamd
{
  extern thread_local std::string g_last_error;

  struct NoDeviceError : std::runtime_error
  {
    using std::runtime_error::runtime_error;
  };
} // namespace mgamd

struct mgamd_tria
{
  std::shared_ptr<mgamd::Tria> tria;
};

struct mgamd_dofs
{
  std::shared_ptr<mgamd::Tria>        tria; // keeps the mesh alive
  std::shared_ptr<mgamd::LevelTables> tables;
  // distributed levels only
  std::shared_ptr<std::vector<uint8_t>>         owned;
  std::shared_ptr<std::map<uint64_t, mgamd::SharedInfo>> shared;
  std::shared_ptr<mgamd::HaloPlan>              halo;
  int                                           n_ranks = 1, rank = 0;
};

struct mgamd_partition
{
  std::vector<std::shared_ptr<mgamd::Tria>> trias; // coarse -> fine
  mgamd::Partition                          part;
};

#define MGAMD_TRY \
  try             \
    {
#define MGAMD_CATCH                              \
  return MGAMD_OK;                               \
  }                                              \
  catch (const mgamd::NoDeviceError &e)          \
  {                                              \
    mgamd::g_last_error = e.what();              \
    return MGAMD_ERR_NO_DEVICE;                  \
  }                                              \
  catch (const std::invalid_argument &e)         \
  {                                              \
    mgamd::g_last_error = e.what();              \
    return MGAMD_ERR_INVALID;                    \
  }                                              \
  catch (const std::exception &e)                \
  {                                              \
    mgamd::g_last_error = e.what();              \
    return MGAMD_ERR;                            \
  }                                              \
  catch (...)                                    \
  {                                              \
    mgamd::g_last_error = "unknown exception";   \
    return MGAMD_ERR;                            \
  }
