#pragma once
#include "../../include/goslam_neus.h"
