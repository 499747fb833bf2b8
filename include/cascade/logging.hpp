// include/cascade/logging.hpp -- counterpart of the reference's include/cascade/logging.hpp:17-24.
//
// The reference logs through an spdlog logger named "cascade" (src/detail/logging_impl.cpp:21-38,
// stdout, default level info) and exposes six level setters (src/logging.cpp:15-48). spdlog is not
// a dependency here: the same six setters drive a small logger that writes spdlog's default line
// format to stdout. At trace level step() emits the reference's per-phase lines with the names its
// users tune the collisional timestep from (src/sim_propagate.cpp:510,526,535,1123,1340-1342,1502,1698;
// src/sim_bvh.cpp:491; src/sim_broad_phase.cpp:209-213; src/sim_narrow_phase.cpp:1017-1021); the
// phases are then timed on the host with a stream synchronisation at each boundary.
#ifndef CASCADE_LOGGING_HPP
#define CASCADE_LOGGING_HPP

namespace cascade
{

void set_logger_level_trace();
void set_logger_level_debug();
void set_logger_level_info();
void set_logger_level_warn();
void set_logger_level_err();
void set_logger_level_critical();

} // namespace cascade

#endif
