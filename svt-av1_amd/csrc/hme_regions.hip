// hme_regions.hip - the level-1/2 HME kernel that does the four regions of a reference in one work item
// (hme_l12_regions_kernel): hme_items.hip's search functions and that kernel, compiled as a unit of their own.
#define SVTME_HME_REGIONS_TU
#include "hme_items.hip"
